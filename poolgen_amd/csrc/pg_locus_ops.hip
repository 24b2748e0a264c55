// pg_locus_ops.hip -- the sync-derived per-locus operators on batches of parsed loci:
//   ols_iter      gwas::ols_iterate   (gwas/ols.rs:201-276)
//   pearson_corr  gwas::correlation   (gwas/correlation_test.rs:73-129, :7-71)
//   chisq_test    tables::chisq       (tables/chisq_test.rs:5-47)
// all of which start with LocusCounts::filter + to_frequencies (base/sync.rs:195-303, :166-192).
//
// Input: counts[L][n][6] u32 (sync columns A,T,C,G,N,D), 24n bytes per locus -- integer/byte
// work, HBM-bound.  ONE LANE PER LOCUS, like the sweep kernel: the filter's pool-size weighted
// allele frequency q_j = sum_i f_ij * w_i must be accumulated sequentially over pools in pool
// order with separate multiply and add (this file is compiled with -ffp-contract=off) to decide
// q < maf exactly like the reference, and a lane walking its own locus does exactly that.
// Coalescing comes from a wave-private LDS transposition: per stage 64 loci x 8 pools (192 B
// each) are fetched as contiguous 16-byte (or 8-byte when n is odd) pieces and written to a
// 208-byte-pitch tile which every lane then reads row-wise.
// Two passes over a locus' counts: pass 1 = coverage + q_j (which alleles survive), pass 2 =
// frequencies over the SURVIVING alleles (row sums change) and the operator's sums.  The
// regression / correlation / chi-square arithmetic closes per lane from those sums.
#include "pg_common.h"
#include "pg_stats_device.h"
#include <cmath>
#include <vector>

namespace {

#ifndef LO_POOL_UNROLL
#define LO_POOL_UNROLL 1
#endif
constexpr int LO_THREADS = 256;
constexpr int LO_WAVES = 4;
constexpr int LO_CHP = 8;                 // pools per stage
constexpr int LO_ROWB = LO_CHP * 24;      // 192 bytes of counts per locus per stage
constexpr int LO_PITCH = LO_ROWB + 16;    // 208: odd number of 16-byte slots
constexpr int LO_TILEB = 64 * LO_PITCH;   // bytes per wave
constexpr int NA = 6;                     // sync alleles
constexpr int MAXK = 2;                   // traits per launch (the host loops over trait pairs)

enum { OP_OLS = 0, OP_PEARSON = 1, OP_CHISQ = 2 };

struct LocusParams {
    int64_t L;
    int n, k;        // k = traits handled by this launch (<= MAXK)
    int k_total, t0; // output layout: trait t0 + tt of k_total
    int remove_ns;
    double min_cov, maf, max_miss;
    int tdf, ntcoef;     // t-test degrees of freedom (OLS: n-1, Pearson: n-2)
    double syy[MAXK];    // OLS: sum of centred y^2
    double sy[MAXK];     // OLS: sum of centred y (~0)
};

typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

// Stage pools [pool0, pool0 + np) of loci l0..l0+63 into the wave's tile.  A locus row of a stage is
// 192 bytes = 12 pieces of 16 B (24 of 8 B when n is odd and rows are only 8-byte aligned): 48 lanes
// cover 4 (2) loci per instruction, so that every address is "lane base + r * constant" -- no
// per-piece address registers -- at the price of 16 idle lanes (loads are not the bottleneck, bytes
// in flight are).  The last, partial stage always moves 8-byte pieces whose offset is clamped
// INSIDE the row (no read past the end of the batch) and which land at that same offset in the tile.
template <int PB, bool FULL>
__device__ __forceinline__ void stage_counts(const uint32_t *__restrict__ counts, char *tile,
                                             int64_t l0, int64_t L, int n, int pool0, int np,
                                             int lane) {
    constexpr int PBE = FULL ? PB : 8;
    constexpr int PPR = LO_ROWB / PBE;     // pieces per locus row: 12 or 24
    constexpr int LPI = 48 / PPR;          // loci per wave instruction: 4 or 2
    constexpr int NI = 64 / LPI;           // instructions per stage: 16 or 32
    if (lane < 48) {
        const int64_t rowb = (int64_t)n * 24;
        const int sub = lane / PPR;
        const int pc = lane - sub * PPR;
        int off = pc * PBE;
        if (!FULL) {
            const int valid = np * 24;
            off = off < valid ? off : valid - PBE;
        }
        const char *gbase = reinterpret_cast<const char *>(counts) + (int64_t)pool0 * 24 + off;
        char *tbase = tile + sub * LO_PITCH + off;
        // all loads of the stage are issued before the first LDS write (NI x 16 bytes in flight per lane)
        if (PBE == 16) {
            uint4_t v[NI];
#pragma unroll
            for (int r = 0; r < NI; ++r) {
                int64_t l = l0 + LPI * r + sub;
                l = l < L ? l : L - 1;
                v[r] = *reinterpret_cast<const uint4_t *>(gbase + l * rowb);
            }
#pragma unroll
            for (int r = 0; r < NI; ++r) *reinterpret_cast<uint4_t *>(tbase + r * (LPI * LO_PITCH)) = v[r];
        } else {
            uint2_t v[NI];
#pragma unroll
            for (int r = 0; r < NI; ++r) {
                int64_t l = l0 + LPI * r + sub;
                l = l < L ? l : L - 1;
                v[r] = *reinterpret_cast<const uint2_t *>(gbase + l * rowb);
            }
#pragma unroll
            for (int r = 0; r < NI; ++r) *reinterpret_cast<uint2_t *>(tbase + r * (LPI * LO_PITCH)) = v[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void read_pool(const char *row, int i, uint32_t (&c)[NA]) {
    const uint2_t a = *reinterpret_cast<const uint2_t *>(row + i * 24);
    const uint2_t b = *reinterpret_cast<const uint2_t *>(row + i * 24 + 8);
    const uint2_t d = *reinterpret_cast<const uint2_t *>(row + i * 24 + 16);
    c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = d.x; c[5] = d.y;
}

// IEEE-correct c / rs for a whole pool from ONE reciprocal.  This is the arithmetic hipcc itself
// emits for an fp64 division of normal-range operands (v_rcp_f64, two Newton steps on the
// reciprocal, q0 = c * r, one fused residual, one fused correction -- the v_div_scale / v_div_fixup
// wrappers only act on over-/underflowing operands, which counts and coverages are not), with the
// reciprocal shared by the (up to six) alleles of the pool instead of being recomputed per allele.
// The quotients are therefore bit-identical to `c / rs`; tests/test_gpu_locus_ops.py pins that
// through the bit-exact mean frequencies and filter decisions.
__device__ __forceinline__ double recip_for_div(double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double r1 = fma(fma(-b, r0, 1.0), r0, r0);
    return fma(fma(-b, r1, 1.0), r1, r1);
}
__device__ __forceinline__ double div_by(double a, double b, double r) {
    const double q0 = a * r;
    return fma(fma(-b, q0, a), r, q0);
}

template <typename T>
__device__ __forceinline__ T pick6(const T (&a)[NA], int idx) {
    T r = a[0];
#pragma unroll
    for (int j = 1; j < NA; ++j) r = (idx == j) ? a[j] : r;
    return r;
}

// index into the packed upper triangle of the 6 x 6 product-sum table
__device__ __forceinline__ constexpr int tri(int a, int b) { // a <= b
    return a * NA - a * (a - 1) / 2 + (b - a);
}

// record layout (struct-of-arrays: field f of locus l at rec[f * L + l])
constexpr int R_CS = 0, R_XX = R_CS + NA, R_XY = R_XX + 21, R_PX = R_XY + NA * MAXK, R_PXX = R_PX + NA * MAXK,
              R_PY = R_PXX + NA * MAXK, R_PYY = R_PY + MAXK, R_PN = R_PYY + MAXK, R_TOTAL = R_PN + MAXK,
              REC_DOUBLES = R_TOTAL + 1;

template <int OP, int PB>
__global__ __launch_bounds__(LO_THREADS, 2) void k_locus_ops(
    const uint32_t *__restrict__ counts, const double *__restrict__ w, const double *__restrict__ Y,
    const double *__restrict__ tcoef, int32_t *__restrict__ n_out, int32_t *__restrict__ ids_out,
    double *__restrict__ mf_out, double *__restrict__ stat_out, double *__restrict__ pv_out,
    int32_t *__restrict__ rec_flags, double *__restrict__ rec, const LocusParams P) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    char *tile = lds_raw + wave * LO_TILEB;
    const char *row = tile + lane * LO_PITCH;
    const int n = P.n, k = P.k;
    const int nst = (n + LO_CHP - 1) / LO_CHP;
    const int64_t ntiles = (P.L + 63) / 64;
    const int64_t wstride = (int64_t)gridDim.x * LO_WAVES;

    for (int64_t t = (int64_t)blockIdx.x * LO_WAVES + wave; t < ntiles; t += wstride) {
        const int64_t l0 = t * 64;
        const int64_t l = l0 + lane;
        // ================= streaming passes =====================================================
        // Pass 1 computes the filter quantities (coverage, q_j) AND, speculatively, the operator's sums
        // with every candidate allele kept.  If every allele the filter then drops has zero counts in
        // every pool (the common case: absent alleles), the row sums over the surviving alleles equal
        // the ones used, i.e. the speculative sums ARE the reference's -- one read of the counts.
        // Otherwise (a dropped allele with reads, e.g. a sequencing-error allele below the MAF) the
        // wave runs pass 2 over the surviving alleles, as the reference's second to_frequencies does.
        double q[NA];
        double mincov = 0.0;
        int n_missing = 0;
        bool anynz[NA];
        double cs[NA];            // NaN-ignoring column sums (sort key, mean frequency)
        double xx[21];            // OLS: sum f_a f_b (a <= b); CHISQ: xx[tri(j,j)] = sum f_j^2 / rs_i
        double xy[NA * MAXK];     // OLS: sum f_j y_t ; PEARSON: sum x y over complete pairs
        double px[NA * MAXK], pxx[NA * MAXK], py[MAXK], pyy[MAXK], pn[MAXK];
        double total = 0.0;
        double shx[NA]; // Pearson: per-allele shift (first valid frequency) for stable one-pass sums
        bool shset = false;
        bool keep[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) { q[j] = 0.0; anynz[j] = false; keep[j] = !(P.remove_ns && j == 4); }

        bool alive = false;
        int nk = 0;
        for (int pass = 0; pass < 2; ++pass) { // one code copy for both passes keeps the register count down
            const bool first = (pass == 0);
#pragma unroll
            for (int j = 0; j < NA; ++j) { cs[j] = 0.0; shx[j] = 0.0; }
#pragma unroll
            for (int j = 0; j < 21; ++j) xx[j] = 0.0;
#pragma unroll
            for (int j = 0; j < NA * MAXK; ++j) { xy[j] = 0.0; px[j] = 0.0; pxx[j] = 0.0; }
#pragma unroll
            for (int j = 0; j < MAXK; ++j) { py[j] = 0.0; pyy[j] = 0.0; pn[j] = 0.0; }
            total = 0.0;
            shset = false;
            for (int st = 0; st < nst; ++st) {
                const int pool0 = st * LO_CHP;
                const int np = min(LO_CHP, n - pool0);
                if (np == LO_CHP) stage_counts<PB, true>(counts, tile, l0, P.L, n, pool0, np, lane);
                else stage_counts<PB, false>(counts, tile, l0, P.L, n, pool0, np, lane);
#pragma unroll LO_POOL_UNROLL
                for (int i = 0; i < np; ++i) {
                    uint32_t c[NA];
                    read_pool(row, i, c);
                    double rs = 0.0; // row sum over the alleles in play (sync.rs:217-222 / :170-175)
#pragma unroll
                    for (int j = 0; j < NA; ++j) rs = keep[j] ? rs + (double)c[j] : rs;
                    const double rinv = recip_for_div(rs);
                    double f[NA];
#pragma unroll
                    for (int j = 0; j < NA; ++j)
                        f[j] = (rs == 0.0) ? NAN : ((c[j] != 0u && keep[j]) ? div_by((double)c[j], rs, rinv) : 0.0);
                    if (first) {
                        mincov = (pool0 + i == 0 || rs < mincov) ? rs : mincov;
                        n_missing += (rs == 0.0) ? 1 : 0;
                        const double wi = w[pool0 + i];
#pragma unroll
                        for (int j = 0; j < NA; ++j) {
                            // q += f * w_i, NaN frequencies contribute 0 (sync.rs:258-271)
                            q[j] = (c[j] != 0u && rs != 0.0 && keep[j]) ? q[j] + f[j] * wi : q[j];
                            anynz[j] = anynz[j] || (c[j] != 0u);
                        }
                    }
                const bool rowok = rs != 0.0;
#pragma unroll
                for (int j = 0; j < NA; ++j) cs[j] = rowok ? cs[j] + f[j] : cs[j];
                if (OP == OP_OLS) {
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int b = a; b < NA; ++b) xx[tri(a, b)] = xx[tri(a, b)] + f[a] * f[b];
#pragma unroll
                    for (int tt = 0; tt < MAXK; ++tt) {
                        if (tt < k) {
                            const double y = Y[(size_t)(pool0 + i) * k + tt];
#pragma unroll
                            for (int j = 0; j < NA; ++j) xy[j * MAXK + tt] = xy[j * MAXK + tt] + f[j] * y;
                        }
                    }
                } else if (OP == OP_PEARSON) {
                    if (rowok && !shset) {
#pragma unroll
                        for (int j = 0; j < NA; ++j) shx[j] = f[j];
                        shset = true;
                    }
#pragma unroll
                    for (int tt = 0; tt < MAXK; ++tt) {
                        if (tt < k) {
                            const double y = Y[(size_t)(pool0 + i) * k + tt]; // shifted by its first valid value on the host
                            const bool ok = rowok && !isnan(y);              // pairwise complete (correlation_test.rs:22-26)
                            py[tt] = ok ? py[tt] + y : py[tt];
                            pyy[tt] = ok ? fma(y, y, pyy[tt]) : pyy[tt];
                            pn[tt] = ok ? pn[tt] + 1.0 : pn[tt];
#pragma unroll
                            for (int j = 0; j < NA; ++j) {
                                const double x = f[j] - shx[j];
                                const int e = j * MAXK + tt;
                                px[e] = ok ? px[e] + x : px[e];
                                pxx[e] = ok ? fma(x, x, pxx[e]) : pxx[e];
                                xy[e] = ok ? fma(x, y, xy[e]) : xy[e];
                            }
                        }
                    }
                } else { // OP_CHISQ: chi2 = total * (sum_j A_j / cs_j - 1), A_j = sum_i f_ij^2 / rs_i
                    double rsum = 0.0;
#pragma unroll
                    for (int j = 0; j < NA; ++j) rsum = keep[j] ? rsum + f[j] : rsum; // row sum of frequencies (~1)
                    total = total + rsum;
#pragma unroll
                    for (int j = 0; j < NA; ++j) xx[tri(j, j)] = xx[tri(j, j)] + (f[j] * f[j]) / rsum;
                }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!first) break;
            bool dropped_with_reads = false;
            nk = 0;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const bool cand = keep[j];
                keep[j] = cand && !((q[j] < P.maf) | (q[j] > (1.00 - P.maf)));
                nk += keep[j] ? 1 : 0;
                dropped_with_reads = dropped_with_reads || (cand && !keep[j] && anynz[j]);
            }
            alive = !(mincov < P.min_cov);                                             // sync.rs:227
            alive = alive && nk >= 2;                                                  // sync.rs:284
            alive = alive && n_missing != n;                                           // sync.rs:293
            alive = alive && !(((double)n_missing / (double)n) > P.max_miss);          // sync.rs:297
            alive = alive && l < P.L;
            if (!__any(alive && dropped_with_reads)) break; // wave-uniform: the speculative sums stand
        }

        if (l >= P.L) continue;
        // The sums go to a per-locus record (struct-of-arrays, coalesced stores); k_locus_close<OP>
        // finishes the statistic.  Keeping the transcendental / LU code out of this kernel keeps its
        // register count -- and so the number of waves that hide the HBM latency -- small.
        {
            int mask = alive ? 1 : 0;
#pragma unroll
            for (int j = 0; j < NA; ++j) mask |= keep[j] ? (2 << j) : 0;
            rec_flags[l] = mask;
#pragma unroll
            for (int j = 0; j < NA; ++j) rec[(size_t)(R_CS + j) * P.L + l] = cs[j];
            if (OP == OP_OLS || OP == OP_CHISQ) {
#pragma unroll
                for (int j = 0; j < 21; ++j)
                    if (OP == OP_OLS || j == tri(0, 0) || j == tri(1, 1) || j == tri(2, 2) || j == tri(3, 3) ||
                        j == tri(4, 4) || j == tri(5, 5))
                        rec[(size_t)(R_XX + j) * P.L + l] = xx[j];
            }
            if (OP == OP_OLS || OP == OP_PEARSON) {
#pragma unroll
                for (int j = 0; j < NA * MAXK; ++j) rec[(size_t)(R_XY + j) * P.L + l] = xy[j];
            }
            if (OP == OP_PEARSON) {
#pragma unroll
                for (int j = 0; j < NA * MAXK; ++j) {
                    rec[(size_t)(R_PX + j) * P.L + l] = px[j];
                    rec[(size_t)(R_PXX + j) * P.L + l] = pxx[j];
                }
#pragma unroll
                for (int j = 0; j < MAXK; ++j) {
                    rec[(size_t)(R_PY + j) * P.L + l] = py[j];
                    rec[(size_t)(R_PYY + j) * P.L + l] = pyy[j];
                    rec[(size_t)(R_PN + j) * P.L + l] = pn[j];
                }
            }
            if (OP == OP_CHISQ) rec[(size_t)R_TOTAL * P.L + l] = total;
        }
    }
}

// Closing kernels: one thread per locus, from the record the streaming kernel left.
template <int OP>
__global__ __launch_bounds__(64) void k_locus_close(const int32_t *__restrict__ rec_flags,
                                                    const double *__restrict__ rec,
                                                    const double *__restrict__ tcoef, int32_t *__restrict__ n_out,
                                                    int32_t *__restrict__ ids_out, double *__restrict__ mf_out,
                                                    double *__restrict__ stat_out, double *__restrict__ pv_out,
                                                    const LocusParams P) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= P.L) return;
    const int n = P.n, k = P.k;
    const int mask = rec_flags[l];
    const bool alive = (mask & 1) != 0;
    bool keep[NA];
    int nk = 0;
    double cs[NA], xx[21], xy[NA * MAXK], px[NA * MAXK], pxx[NA * MAXK], py[MAXK], pyy[MAXK], pn[MAXK];
    double total = 0.0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        keep[j] = (mask & (2 << j)) != 0;
        nk += keep[j] ? 1 : 0;
        cs[j] = rec[(size_t)(R_CS + j) * P.L + l];
    }
#pragma unroll
    for (int j = 0; j < 21; ++j) xx[j] = 0.0;
    if (OP == OP_OLS) {
#pragma unroll
        for (int j = 0; j < 21; ++j) xx[j] = rec[(size_t)(R_XX + j) * P.L + l];
    } else if (OP == OP_CHISQ) {
#pragma unroll
        for (int j = 0; j < NA; ++j) xx[tri(j, j)] = rec[(size_t)(R_XX + tri(j, j)) * P.L + l];
        total = rec[(size_t)R_TOTAL * P.L + l];
    }
#pragma unroll
    for (int j = 0; j < NA * MAXK; ++j) {
        xy[j] = (OP == OP_OLS || OP == OP_PEARSON) ? rec[(size_t)(R_XY + j) * P.L + l] : 0.0;
        px[j] = (OP == OP_PEARSON) ? rec[(size_t)(R_PX + j) * P.L + l] : 0.0;
        pxx[j] = (OP == OP_PEARSON) ? rec[(size_t)(R_PXX + j) * P.L + l] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        py[j] = (OP == OP_PEARSON) ? rec[(size_t)(R_PY + j) * P.L + l] : 0.0;
        pyy[j] = (OP == OP_PEARSON) ? rec[(size_t)(R_PYY + j) * P.L + l] : 0.0;
        pn[j] = (OP == OP_PEARSON) ? rec[(size_t)(R_PN + j) * P.L + l] : 0.0;
    }
        // ================= closing arithmetic per locus ==========================================
        if (OP == OP_CHISQ) {
            // tables/chisq_test.rs:15-35 on the frequency table of the surviving alleles
            int cnt = 0;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (keep[j]) {
                    ids_out[l * PG_MAX_OUT + (cnt < PG_MAX_OUT ? cnt : PG_MAX_OUT - 1)] = j;
                    acc += xx[tri(j, j)] / cs[j];
                    ++cnt;
                }
            }
            const double chi2 = total * (acc - 1.0);
            const double df = (double)(n * nk) - 1.0;
            n_out[l] = alive ? nk : 0;
            stat_out[l] = alive ? chi2 : NAN;
            pv_out[l] = alive ? pg_chisq_upper_p(chi2, df, pg_ln_gamma(df / 2.0)) : NAN;
            return;
        }

        // order of the surviving alleles
        int ord[NA]; // ord[r] = allele id at rank r (only the first nk entries are meaningful)
        if (OP == OP_OLS) {
            // stable sort by decreasing column sum (sync.rs:477-506), then drop rank 0 (ols.rs:227-230)
            int rank[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                int r = 0;
#pragma unroll
                for (int i2 = 0; i2 < NA; ++i2)
                    if (i2 != j && keep[i2]) r += (cs[i2] > cs[j] || (cs[i2] == cs[j] && i2 < j)) ? 1 : 0;
                rank[j] = keep[j] ? r : NA;
            }
#pragma unroll
            for (int r = 0; r < NA; ++r) {
                int id = 0;
#pragma unroll
                for (int j = 0; j < NA; ++j) id = (rank[j] == r) ? j : id;
                ord[r] = id;
            }
        } else {
            int r = 0;
#pragma unroll
            for (int j = 0; j < NA; ++j) ord[j] = 0;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (keep[j]) {
#pragma unroll
                    for (int s = 0; s < NA; ++s) ord[s] = (s == r) ? j : ord[s];
                    ++r;
                }
            }
        }

        if (OP == OP_PEARSON) {
            // gwas/correlation_test.rs:94-126: all surviving alleles but the LAST, unsorted
            const int nout = nk >= 2 ? nk - 1 : nk;
            n_out[l] = alive ? nout : 0;
#pragma unroll
            for (int r = 0; r < PG_MAX_OUT; ++r) {
                const int j = ord[r];
                const bool on = alive && r < nout;
                ids_out[l * PG_MAX_OUT + r] = on ? j : -1;
                mf_out[l * PG_MAX_OUT + r] = on ? pick6(cs, j) / (double)n : NAN; // x.mean(), :119
#pragma unroll
                for (int tt = 0; tt < MAXK; ++tt) {
                    if (tt >= k) continue;
                    double rr = NAN, pp = NAN;
                    if (on) {
                        double sx = 0, sxx = 0, sxy = 0;
                        const double sy = py[tt], syy = pyy[tt], m = pn[tt];
#pragma unroll
                        for (int jj = 0; jj < NA; ++jj) {
                            const int e = jj * MAXK + tt;
                            sx = (jj == j) ? px[e] : sx; sxx = (jj == j) ? pxx[e] : sxx;
                            sxy = (jj == j) ? xy[e] : sxy;
                        }
                        const double cxy = sxy - sx * sy / m;
                        const double cxx = sxx - sx * sx / m;
                        const double cyy = syy - sy * sy / m;
                        const double r0 = cxy / (sqrt(cxx) * sqrt(cyy));      // :50-52
                        if (isnan(r0)) { rr = NAN; pp = NAN; }                // :53-56
                        else {
                            const double sden = (1.0 - r0 * r0) / ((double)n - 2.0); // :57
                            if (sden <= 0.0) { rr = r0; pp = PG_EPS; }         // :58-61
                            else {
                                const double tstat = r0 / sqrt(sden);
                                pp = (n > 2) ? pg_t_two_sided_p(fabs(tstat), P.tdf, tcoef, P.ntcoef) : NAN;
                                rr = round(r0 * 1e7) / 1e7; // sensible_round(r, 7), :70 (half away from zero)
                            }
                        }
                    }
                    stat_out[(l * PG_MAX_OUT + r) * P.k_total + P.t0 + tt] = rr;
                    pv_out[(l * PG_MAX_OUT + r) * P.k_total + P.t0 + tt] = pp;
                }
            }
            return;
        }

        // ---------------- OP_OLS: literal normal equations in the reference's column order -------
        // X = [1 | f_ord[1] | ... | f_ord[nk-1]]  (ols.rs:240-246), P = nk columns
        const int Pn = nk;
        double A[NA][NA];
        int sel[NA]; // sel[r] = allele id of design column r (r >= 1), -1 for the intercept / unused
#pragma unroll
        for (int r = 0; r < NA; ++r) sel[r] = (r >= 1 && r < Pn) ? ord[r] : -1;
        // entry (r, c) of X'X, identity-padded beyond P so that the padding is inert in the LU
        auto xtx = [&](int r, int c2) -> double {
            double v;
            if (r >= Pn || c2 >= Pn) v = (r == c2) ? 1.0 : 0.0;
            else if (r == 0 && c2 == 0) v = (double)n;
            else if (r == 0) v = pick6(cs, sel[c2]);
            else if (c2 == 0) v = pick6(cs, sel[r]);
            else {
                const int a = min(sel[r], sel[c2]), bq = max(sel[r], sel[c2]);
                double sacc = 0.0;
#pragma unroll
                for (int i2 = 0; i2 < 21; ++i2) sacc = (i2 == tri(a, bq)) ? xx[i2] : sacc;
                v = sacc;
            }
            return v;
        };
#pragma unroll
        for (int r = 0; r < NA; ++r)
#pragma unroll
            for (int c2 = 0; c2 < NA; ++c2) A[r][c2] = xtx(r, c2);
        // LU with partial pivoting, first max |a| in the column (the oracle's lu_factor; LAPACK dgetf2)
        bool singular = false;
        int piv[NA];
#pragma unroll
        for (int kk = 0; kk < NA; ++kk) {
            int pi = kk;
            double pm = fabs(A[kk][kk]);
#pragma unroll
            for (int i2 = kk + 1; i2 < NA; ++i2) {
                const double v = fabs(A[i2][kk]);
                const bool g = v > pm;
                pm = g ? v : pm;
                pi = g ? i2 : pi;
            }
            piv[kk] = pi;
#pragma unroll
            for (int i2 = kk + 1; i2 < NA; ++i2) {
                const bool sw = (pi == i2);
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const double x1 = A[kk][j], x2 = A[i2][j];
                    A[kk][j] = sw ? x2 : x1;
                    A[i2][j] = sw ? x1 : x2;
                }
            }
            if (A[kk][kk] == 0.0) singular = true;
            const double inv = 1.0 / A[kk][kk];
#pragma unroll
            for (int i2 = kk + 1; i2 < NA; ++i2) A[i2][kk] = A[i2][kk] * inv;
#pragma unroll
            for (int i2 = kk + 1; i2 < NA; ++i2) {
                const double lf = A[i2][kk];
#pragma unroll
                for (int j = kk + 1; j < NA; ++j) A[i2][j] = A[i2][j] - lf * A[kk][j];
            }
        }
        // x = (X'X)^-1 rhs through the factorisation (P A = L U)
        auto lu_solve = [&](double (&col)[NA]) {
#pragma unroll
            for (int kk = 0; kk < NA; ++kk) {
#pragma unroll
                for (int i2 = kk + 1; i2 < NA; ++i2) {
                    const bool sw = (piv[kk] == i2);
                    const double x1 = col[kk], x2 = col[i2];
                    col[kk] = sw ? x2 : x1;
                    col[i2] = sw ? x1 : x2;
                }
            }
#pragma unroll
            for (int i2 = 0; i2 < NA; ++i2) {
                double sacc = col[i2];
#pragma unroll
                for (int j = 0; j < i2; ++j) sacc = sacc - A[i2][j] * col[j];
                col[i2] = sacc;
            }
#pragma unroll
            for (int i2 = NA - 1; i2 >= 0; --i2) {
                double sacc = col[i2];
#pragma unroll
                for (int j = i2 + 1; j < NA; ++j) sacc = sacc - A[i2][j] * col[j];
                col[i2] = sacc / A[i2][i2];
            }
        };
        // full inverse, column by column in the oracle's order: its diagonal gives var(b)
        // (ols.rs:111-116) and its determinant feeds the second singularity test (ols.rs:81-83)
        double Inv[NA][NA], dinv[NA];
#pragma unroll
        for (int c2 = 0; c2 < NA; ++c2) {
            double col[NA];
#pragma unroll
            for (int i2 = 0; i2 < NA; ++i2) col[i2] = (i2 == c2) ? 1.0 : 0.0;
            lu_solve(col);
#pragma unroll
            for (int i2 = 0; i2 < NA; ++i2) Inv[i2][c2] = col[i2];
            dinv[c2] = col[c2];
        }
        // `inv.det() == 0.0` (ols.rs:81): LU of the inverse, singular factorisation -> det 0.
        // Loci with duplicated allele columns pass the first LU by a rounding residue and are
        // caught here, exactly as in the reference.
        {
            double det = 1.0;
            bool zero_piv = false;
#pragma unroll
            for (int kk = 0; kk < NA; ++kk) {
                int pi = kk;
                double pm = fabs(Inv[kk][kk]);
#pragma unroll
                for (int i2 = kk + 1; i2 < NA; ++i2) {
                    const double v = fabs(Inv[i2][kk]);
                    const bool g = v > pm;
                    pm = g ? v : pm;
                    pi = g ? i2 : pi;
                }
#pragma unroll
                for (int i2 = kk + 1; i2 < NA; ++i2) {
                    const bool sw = (pi == i2);
#pragma unroll
                    for (int j = 0; j < NA; ++j) {
                        const double x1 = Inv[kk][j], x2 = Inv[i2][j];
                        Inv[kk][j] = sw ? x2 : x1;
                        Inv[i2][j] = sw ? x1 : x2;
                    }
                }
                const double pvt = Inv[kk][kk];
                zero_piv = zero_piv || (pvt == 0.0);
                const double ipv = 1.0 / pvt;
#pragma unroll
                for (int i2 = kk + 1; i2 < NA; ++i2) {
                    const double lf = (pvt != 0.0) ? Inv[i2][kk] * ipv : 0.0;
                    if (lf != 0.0) {
#pragma unroll
                        for (int j = kk + 1; j < NA; ++j) Inv[i2][j] = Inv[i2][j] - lf * Inv[kk][j];
                    }
                }
                det = det * pvt;
            }
            if (zero_piv || det == 0.0) singular = true;
        }
        const bool ok = alive && !singular; // Err -> the whole locus is dropped (ols.rs:250-253)
        if (P.t0 == 0) {
            n_out[l] = ok ? Pn - 1 : 0;
#pragma unroll
            for (int r = 1; r < NA; ++r) {
                const bool on = ok && r < Pn;
                if (r - 1 < PG_MAX_OUT) {
                    ids_out[l * PG_MAX_OUT + r - 1] = on ? sel[r] : -1;
                    mf_out[l * PG_MAX_OUT + r - 1] = on ? pick6(cs, sel[r]) / (double)n : NAN; // ols.rs:266
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < MAXK; ++tt) {
            if (tt >= k) continue;
            // X'y with the centred phenotype (slopes are invariant to the shift; it removes the
            // y-bar^2 cancellation from the residual sum of squares)
            double xty[NA], b[NA];
#pragma unroll
            for (int r = 0; r < NA; ++r) {
                double v = 0.0;
                if (r == 0) v = P.sy[tt];
                else {
#pragma unroll
                    for (int j = 0; j < NA; ++j) v = (sel[r] == j) ? xy[j * MAXK + tt] : v;
                }
                xty[r] = (r < Pn) ? v : 0.0;
                b[r] = xty[r];
            }
            lu_solve(b);
            // RSS = y'y - 2 b'X'y + b'(X'X) b: the form that is stationary in b, so the O(cond*eps)
            // error of the solve enters only to second order
            double bxy = 0.0;
#pragma unroll
            for (int r = 0; r < NA; ++r) {
                double ab = 0.0;
#pragma unroll
                for (int c2 = 0; c2 < NA; ++c2) ab = (r < Pn && c2 < Pn) ? fma(xtx(r, c2), b[c2], ab) : ab;
                bxy = (r < Pn) ? fma(b[r], 2.0 * xty[r] - ab, bxy) : bxy;
            }
            double rss = P.syy[tt] - bxy;
            rss = rss < 0.0 ? 0.0 : rss;
            const double ve = rss / ((double)n - (double)Pn); // ols.rs:103
#pragma unroll
            for (int r = 1; r < NA; ++r) {
                if (r - 1 >= PG_MAX_OUT) continue;
                const bool on = ok && r < Pn;
                double pv = NAN, bb = NAN;
                if (on) {
                    bb = b[r];
                    const double vb = ve * dinv[r];                                  // ols.rs:111-116
                    const double tstat = (fabs(bb) <= PG_EPS) ? 0.0 : bb / sqrt(vb); // ols.rs:143-147
                    if (fabs(tstat) <= PG_EPS) pv = 1.0;
                    else if (isnan(tstat)) pv = 1.0;
                    else pv = pg_t_two_sided_p(fabs(tstat), P.tdf, tcoef, P.ntcoef);
                }
                stat_out[(l * PG_MAX_OUT + r - 1) * P.k_total + P.t0 + tt] = bb;
                pv_out[(l * PG_MAX_OUT + r - 1) * P.k_total + P.t0 + tt] = pv;
            }
        }
}

// ---------------------------------------------------------------------------------------------
template <int OP>
int run_locus_op(pg_ctx *ctx, int kid, const uint32_t *counts_dev, int64_t L, int n,
                 const double *pool_sizes, const pg_filter *flt, const double *Y, int k,
                 int32_t *n_out, int32_t *ids, double *mf, double *stat, double *pv) {
    PG_CHECK(ctx, counts_dev && pool_sizes && flt && n_out && ids && stat && pv, "locus op: null pointer");
    PG_CHECK(ctx, L > 0 && n >= 1, "locus op: bad shape L=%lld n=%d", (long long)L, n);
    PG_CHECK(ctx, OP == OP_CHISQ || (Y && k >= 1), "locus op: need at least one trait");
    PG_CHECK(ctx, (reinterpret_cast<uintptr_t>(counts_dev) & 15) == 0, "locus op: counts must be 16-byte aligned");
    PG_CHECK(ctx, OP != OP_OLS || n >= 2, "ols_iter: StudentsT needs n - 1 >= 1 degrees of freedom");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    if (OP == OP_CHISQ) k = 1;
    if (OP == OP_OLS)
        for (int i = 0; i < n * k; ++i)
            PG_CHECK(ctx, !std::isnan(Y[i]), "ols_iter: remove pools with missing phenotypes first "
                                             "(remove_missing, gwas/ols.rs:206)");
    // pool weights exactly as the reference forms them: pool_sizes[i] / sum(pool_sizes) (sync.rs:266-268)
    std::vector<double> w(n);
    double total = 0.0;
    for (int i = 0; i < n; ++i) total = total + pool_sizes[i];
    for (int i = 0; i < n; ++i) w[i] = pool_sizes[i] / total;
    const int df = (OP == OP_OLS) ? n - 1 : n - 2; // ols.rs:139 / correlation_test.rs:65
    std::vector<double> tc = pg_tdist_coef(df < 1 ? 1 : df);
    const size_t side = ((size_t)n + (size_t)n * MAXK + tc.size() + 8 + 1) & ~(size_t)1; // doubles, even
    const size_t recd = (size_t)L * REC_DOUBLES;
    const size_t need = sizeof(double) * (side + recd) + sizeof(int32_t) * (size_t)L;
    int rc = pg_ws_reserve(ctx, need);
    if (rc) return rc;
    double *wd = static_cast<double *>(ctx->ws);
    double *Ydev = wd + n;
    double *tcd = Ydev + (size_t)n * MAXK;
    double *recp = wd + side;
    int32_t *recf = reinterpret_cast<int32_t *>(recp + recd);
    PG_HIP(ctx, hipMemcpyAsync(wd, w.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (!tc.empty())
        PG_HIP(ctx, hipMemcpyAsync(tcd, tc.data(), sizeof(double) * tc.size(), hipMemcpyHostToDevice, ctx->stream));
    const int cus = ctx->cus;
    const int64_t ntiles = (L + 63) / 64;
    int64_t blocks = (ntiles + LO_WAVES - 1) / LO_WAVES;
    const int64_t cap = (int64_t)cus * 8;
    const int grid = (int)(blocks < cap ? blocks : cap);
    const size_t shmem = (size_t)LO_WAVES * LO_TILEB;
    const bool p16 = ((int64_t)n * 24) % 16 == 0;
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_locus_ops<OP, 16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_locus_ops<OP, 8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));

    std::vector<double> Yd((size_t)n * MAXK);
    for (int t0 = 0; t0 < k; t0 += MAXK) { // the filter passes are recomputed per trait pair
        const int kg = (k - t0) < MAXK ? (k - t0) : MAXK;
        LocusParams P;
        std::memset(&P, 0, sizeof P);
        P.L = L; P.n = n; P.k = kg; P.k_total = k; P.t0 = t0;
        P.remove_ns = flt->remove_ns ? 1 : 0;
        P.min_cov = (double)flt->min_coverage_depth;
        P.maf = flt->min_allele_frequency;
        P.max_miss = flt->max_missingness_rate;
        P.tdf = df;
        P.ntcoef = (int)tc.size();
        std::fill(Yd.begin(), Yd.end(), 0.0);
        if (OP == OP_OLS) {
            for (int t = 0; t < kg; ++t) {
                double mu = 0.0;
                for (int i = 0; i < n; ++i) mu += Y[(size_t)i * k + t0 + t];
                mu /= n;
                double sy = 0.0, syy = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double y = Y[(size_t)i * k + t0 + t] - mu;
                    Yd[(size_t)i * kg + t] = y;
                    sy += y;
                    syy += y * y;
                }
                P.sy[t] = sy;
                P.syy[t] = syy;
            }
        } else if (OP == OP_PEARSON) {
            for (int t = 0; t < kg; ++t) {
                double sh = 0.0;
                for (int i = 0; i < n; ++i)
                    if (!std::isnan(Y[(size_t)i * k + t0 + t])) { sh = Y[(size_t)i * k + t0 + t]; break; }
                for (int i = 0; i < n; ++i) Yd[(size_t)i * kg + t] = Y[(size_t)i * k + t0 + t] - sh;
            }
        }
        PG_HIP(ctx, hipMemcpyAsync(Ydev, Yd.data(), sizeof(double) * n * MAXK, hipMemcpyHostToDevice, ctx->stream));
        pg_prof_begin(ctx, kid);
        if (p16)
            hipLaunchKernelGGL((k_locus_ops<OP, 16>), dim3(grid), dim3(LO_THREADS), shmem, ctx->stream, counts_dev,
                               wd, Ydev, tcd, n_out, ids, mf, stat, pv, recf, recp, P);
        else
            hipLaunchKernelGGL((k_locus_ops<OP, 8>), dim3(grid), dim3(LO_THREADS), shmem, ctx->stream, counts_dev,
                               wd, Ydev, tcd, n_out, ids, mf, stat, pv, recf, recp, P);
        hipLaunchKernelGGL(k_locus_close<OP>, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, ctx->stream, recf, recp, tcd,
                           n_out, ids, mf, stat, pv, P);
        pg_prof_end(ctx);
        PG_HIP(ctx, hipGetLastError());
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // Yd is reused by the next trait pair
    }
    return PG_OK;
}

// host-buffer wrapper: H2D, run, D2H
template <int OP>
int run_locus_op_host(pg_ctx *ctx, int kid, const uint32_t *counts, int64_t L, int n,
                      const double *pool_sizes, const pg_filter *flt, const double *Y, int k,
                      int32_t *n_out, int32_t *ids, double *mf, double *stat, double *pv) {
    PG_CHECK(ctx, counts && n_out && ids && stat && pv && L > 0 && n >= 1, "locus op: bad arguments");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int kk = (OP == OP_CHISQ) ? 1 : k;
    const size_t cb = (size_t)L * n * 6 * sizeof(uint32_t);
    const size_t sb = (OP == OP_CHISQ) ? (size_t)L * sizeof(double) : (size_t)L * PG_MAX_OUT * kk * sizeof(double);
    const size_t ib = (size_t)L * PG_MAX_OUT * sizeof(int32_t), mb = (size_t)L * PG_MAX_OUT * sizeof(double);
    char *d = nullptr;
    PG_HIP(ctx, hipMalloc((void **)&d, cb + 2 * sb + ib + mb + (size_t)L * 4 + 256));
    uint32_t *cd = reinterpret_cast<uint32_t *>(d);
    double *sd = reinterpret_cast<double *>(d + ((cb + 15) & ~(size_t)15));
    double *pd = sd + sb / 8;
    double *md = pd + sb / 8;
    int32_t *idd = reinterpret_cast<int32_t *>(md + mb / 8);
    int32_t *nd = idd + ib / 4;
    int rc = PG_OK;
    if (hipMemcpyAsync(cd, counts, cb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = pg_fail(ctx, PG_ERR_HIP, "locus op: H2D failed");
    if (!rc) rc = run_locus_op<OP>(ctx, kid, cd, L, n, pool_sizes, flt, Y, k, nd, idd, md, sd, pd);
    if (!rc) {
        bool okc = hipMemcpyAsync(n_out, nd, (size_t)L * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        okc = okc && hipMemcpyAsync(ids, idd, ib, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        if (mf) okc = okc && hipMemcpyAsync(mf, md, mb, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        okc = okc && hipMemcpyAsync(stat, sd, sb, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        okc = okc && hipMemcpyAsync(pv, pd, sb, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        if (!okc) rc = pg_fail(ctx, PG_ERR_HIP, "locus op: D2H failed");
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

} // namespace

extern "C" int pg_ols_iter_batch_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n,
                                     const double *pool_sizes, const pg_filter *filter, const double *Y,
                                     int k, int32_t *n_out_dev, int32_t *allele_ids_dev,
                                     double *mean_freq_dev, double *stat_dev, double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, mean_freq_dev, "ols_iter: null mean_freq");
    return run_locus_op<OP_OLS>(ctx, PG_K_OLS_ITER, counts_dev, L, n, pool_sizes, filter, Y, k, n_out_dev,
                                allele_ids_dev, mean_freq_dev, stat_dev, pval_dev);
}

extern "C" int pg_pearson_batch_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n,
                                    const double *pool_sizes, const pg_filter *filter, const double *Y,
                                    int k, int32_t *n_out_dev, int32_t *allele_ids_dev,
                                    double *mean_freq_dev, double *stat_dev, double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, mean_freq_dev, "pearson: null mean_freq");
    return run_locus_op<OP_PEARSON>(ctx, PG_K_PEARSON, counts_dev, L, n, pool_sizes, filter, Y, k, n_out_dev,
                                    allele_ids_dev, mean_freq_dev, stat_dev, pval_dev);
}

extern "C" int pg_chisq_batch_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n,
                                  const double *pool_sizes, const pg_filter *filter, int32_t *n_out_dev,
                                  int32_t *allele_ids_dev, double *chi2_dev, double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    return run_locus_op<OP_CHISQ>(ctx, PG_K_CHISQ, counts_dev, L, n, pool_sizes, filter, nullptr, 1, n_out_dev,
                                  allele_ids_dev, nullptr, chi2_dev, pval_dev);
}

extern "C" int pg_ols_iter_batch(pg_ctx *ctx, const uint32_t *counts, int64_t L, int n,
                                 const double *pool_sizes, const pg_filter *filter, const double *Y, int k,
                                 int32_t *n_out, int32_t *allele_ids, double *mean_freq, double *stat,
                                 double *pval) {
    if (!ctx) return PG_ERR_INVALID;
    return run_locus_op_host<OP_OLS>(ctx, PG_K_OLS_ITER, counts, L, n, pool_sizes, filter, Y, k, n_out,
                                     allele_ids, mean_freq, stat, pval);
}

extern "C" int pg_pearson_batch(pg_ctx *ctx, const uint32_t *counts, int64_t L, int n,
                                const double *pool_sizes, const pg_filter *filter, const double *Y, int k,
                                int32_t *n_out, int32_t *allele_ids, double *mean_freq, double *stat,
                                double *pval) {
    if (!ctx) return PG_ERR_INVALID;
    return run_locus_op_host<OP_PEARSON>(ctx, PG_K_PEARSON, counts, L, n, pool_sizes, filter, Y, k, n_out,
                                         allele_ids, mean_freq, stat, pval);
}

extern "C" int pg_chisq_batch(pg_ctx *ctx, const uint32_t *counts, int64_t L, int n,
                              const double *pool_sizes, const pg_filter *filter, int32_t *n_out,
                              int32_t *allele_ids, double *chi2, double *pval) {
    if (!ctx) return PG_ERR_INVALID;
    return run_locus_op_host<OP_CHISQ>(ctx, PG_K_CHISQ, counts, L, n, pool_sizes, filter, nullptr, 1, n_out,
                                       allele_ids, nullptr, chi2, pval);
}
