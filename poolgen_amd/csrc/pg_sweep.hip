// pg_sweep.hip -- the per-locus OLS sweep of ols_iter_with_kinship and its host-side set-up.
//
// Reference (gwas/ols.rs:340-370): for every column g of G and every trait y,
//     X = [1 | C | g],  b = (X^T X)^-1 X^T y,  report b, var(b), p of the LAST coefficient
// with the fit of gwas/ols.rs:58-160 (ve = e'e/(n-P), t = b/sqrt(var), p = 2(1 - T_{n-1}(|t|))).
// Z = [1 | C] is the same for every locus, so with Q an orthonormal basis of span(Z) and
// ytilde = y - Q Q^T y (host, once) the last coefficient is the Frisch-Waugh-Lovell ratio
//     u = Q^T g,  s_gg = g'g - u'u,  s_gy = g'ytilde,
//     b = s_gy / s_gg,  RSS = ytilde'ytilde - s_gy^2 / s_gg,  var(b) = RSS / (n-P) / s_gg
// which equals the reference's [(X^T X)^-1]_(last,last) formulation up to O(cond * eps).
// Per locus that is (m+1+k) dot products of length n against vectors shared by all loci plus
// g'g: ~0.75-3 flop per byte of G, i.e. HBM-read bound.  Algorithmic traffic: 8n bytes read +
// 24k bytes written per locus.
//
// Kernel design (gfx950): ONE LANE PER LOCUS, so the (m+1+k) running sums live in registers,
// the shared vectors W = [Q | ytilde] are wave-uniform operands fetched through the scalar
// cache, and no cross-lane reduction is ever needed; the closing p-value code runs on all 64
// lanes.  Because a lane walking its own 8n-byte row would be an uncoalesced access, each wave
// transposes through a PRIVATE LDS tile: 16 lanes fetch 256 contiguous bytes (32 pools) of one
// locus per global_load_dwordx4, 64 loci x 32 pools are written to LDS with a 272-byte row
// pitch (odd number of 16-byte slots => conflict-free ds_read_b128 by lane=row), and then every
// lane reads its own row.  Tiles are wave-private: no workgroup barriers at all, and the
// hardware overlaps one wave's loads with another wave's FMAs.
#include "pg_common.h"
#include "pg_stats_device.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

constexpr int SW_THREADS = 256;
constexpr int SW_WAVES = SW_THREADS / 64;
#ifndef SW_CH_DEF
#define SW_CH_DEF 32
#endif
#ifndef SW_MINWAVES
#define SW_MINWAVES 1
#endif
constexpr int SW_CH = SW_CH_DEF;     // pools per LDS chunk (32 or 16)
constexpr int SW_PITCH = SW_CH + 2;  // doubles per LDS row: an odd number of 16-byte slots
constexpr int SW_LPR = SW_CH / 2;    // lanes (16-byte pieces) per locus row of a chunk
constexpr int SW_RPI = 64 / SW_LPR;  // locus rows per wave load instruction
constexpr int SW_NLD = 64 / SW_RPI;  // load instructions per chunk
constexpr int SW_TILE = 64 * SW_PITCH;

struct SweepDims {
    int64_t p, ld;
    int64_t ntiles;
    int n, m1, k, tdf, ntcoef;
    int colmajor; // k_gp_beta*: out is k x p instead of p x k
    double *ss;   // k_gp_beta: if set, sum of squares of every row (the MLE path's g'g)
    double *lz;   // k_ols_sweep_mfma MODE 2: per wave (sum over its loci of (sum_i g_i)^2, sum over its loci of sum_i g_i^2)
    double dfe; // n - P as f64 (ols.rs:103)
    double tau; // relative singularity threshold on s_gg / g'g
};

// Closing arithmetic of one (locus, trait) fit from the projected sums (gwas/ols.rs:102-116, 139-158)
__device__ __forceinline__ void ols_close(double sgg, double sgy, double syy, bool bad, double dfe, int tdf,
                                          const double *__restrict__ tcoef, int ntcoef, double &b,
                                          double &vb, double &pv) {
    b = NAN; vb = NAN; pv = NAN;
    if (bad) return;
    b = sgy / sgg;
    double rss = syy - sgy * b;
    rss = rss < 0.0 ? 0.0 : rss;
    vb = (rss / dfe) / sgg;
    const double tt = (fabs(b) <= PG_EPS) ? 0.0 : b / sqrt(vb);
    if (fabs(tt) <= PG_EPS) pv = 1.0;
    else if (isnan(tt)) pv = 1.0;
    else pv = pg_t_two_sided_p(fabs(tt), tdf, tcoef, ntcoef);
}

// One 32-pool chunk of a 64-locus tile: coalesced global loads -> wave-private LDS tile ->
// lane-per-locus accumulation.  FULL = all 32 pools valid (compile-time trip count).
template <int C, bool FULL, bool SHIFT = true>
__device__ __forceinline__ void sweep_chunk(const double *__restrict__ G,
                                            const double *__restrict__ Wp, double *tile,
                                            int64_t l0, int64_t p, int64_t ld, int pool0, int npool,
                                            int lane, bool first, double &shift, double &s2,
                                            double (&acc)[C]) {
    const int lr = lane / SW_LPR;
    const int piece = lane % SW_LPR;
    // ---- global -> registers: 16 x (4 loci x 256 B), branch-free (clamped address + select)
    int cofs = 2 * piece;
    bool col_ok = true, two = true;
    if (!FULL) {
        col_ok = cofs < npool;
        two = (cofs + 1) < npool;
        const int last = (npool - 1) & ~1;
        cofs = cofs < last ? cofs : last;
    }
    double2 v[SW_NLD];
#pragma unroll
    for (int r = 0; r < SW_NLD; ++r) {
        int64_t l = l0 + SW_RPI * r + lr;
        l = l < p ? l : p - 1;
        v[r] = *reinterpret_cast<const double2 *>(G + l * ld + pool0 + cofs);
    }
#pragma unroll
    for (int r = 0; r < SW_NLD; ++r) {
        double2 x = v[r];
        if (!FULL) {
            x.x = col_ok ? x.x : 0.0;
            x.y = two ? x.y : 0.0;
        }
        *reinterpret_cast<double2 *>(&tile[(SW_RPI * r + lr) * SW_PITCH + 2 * piece]) = x;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- lane = locus: walk the row; W is wave-uniform (scalar-cache operands) -----------------
    const double *row = tile + lane * SW_PITCH;
    if (SHIFT && first) shift = row[0]; // any per-locus constant cancels because Z contains the intercept
    if (FULL) {
#pragma unroll
        for (int i = 0; i < SW_CH; i += 2) {
            const double2 g2 = *reinterpret_cast<const double2 *>(&row[i]);
            const double ga = g2.x - shift;
            const double gb = g2.y - shift;
            s2 = fma(ga, ga, s2);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fma(ga, Wp[i * C + c], acc[c]);
            s2 = fma(gb, gb, s2);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fma(gb, Wp[(i + 1) * C + c], acc[c]);
        }
    } else {
        for (int i = 0; i < npool; ++i) {
            const double ga = row[i] - shift;
            s2 = fma(ga, ga, s2);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fma(ga, Wp[i * C + c], acc[c]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- the sweep proper ---------------------------------------------------------------------------------------
// Memory access: every 128-byte line of G is fetched exactly ONCE.  A row (locus) is 8*ld bytes, so consecutive rows start
// at different offsets inside a line (ld = 200: every other row starts 64 bytes into one) and a per-row grid of 256-byte
// chunks would touch the lines at both ends of a chunk twice -- measured on the first version of this kernel as 1.14x the
// algorithmic bytes (18.49 GB moved for 16.24 GB).  Here g = 128 / gcd(8 ld, 128) consecutive rows (1, 2, 4 or 8) form a
// SUPER-ROW of g*ld doubles that starts and ends on a line boundary; a lane owns one super-row and walks its g loci one
// after the other, a wave tile is 64 super-rows = one contiguous, line-aligned slab, and the slab is read in aligned
// 256-byte chunks (16 lanes x 16 bytes each, 4 super-rows per load instruction).  All lanes cross from one locus to the
// next at the same chunk position, so the pool index -- and with it the W operand -- stays wave-uniform (scalar loads).
// The loads of chunk c + 1 (or of the next tile's first chunk) are in flight while chunk c is consumed and while the
// p-value code of a finished locus runs: 16 KB per wave, 8 waves per CU.
struct SweepGeom {
    int g;          // loci per super-row
    int nch;        // 32-double chunks per super-row
    int64_t sl;     // doubles per super-row = g * ld
    int64_t total;  // p * ld: loads are clamped to the last pair of the matrix
    int prefetch;   // experiments: 0 = the loads of a chunk are issued when the chunk is needed
};

// The 16 staging registers of a chunk are named variables, not an array: they are live across the consumption loop (the
// prefetch), and an array that is live around a loop with this much control flow ends up in scratch memory.
static_assert(SW_NLD == 16, "the staging macros below spell out 16 load instructions per chunk");
#define SW_REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define SW_DECL(r) double2 v##r;
#define SW_LOAD(r)                                                                                        \
    {                                                                                                     \
        int64_t idx_ = ibase_ + (int64_t)(SW_RPI * r) * Q.sl;                                             \
        idx_ = idx_ < lastpair ? idx_ : lastpair;                                                         \
        v##r = *reinterpret_cast<const double2 *>(G + idx_);                                              \
    }
#define SW_ISSUE(tt, cc)                                                                                  \
    {                                                                                                     \
        int64_t q_ = (int64_t)(cc) * SW_CH + 2 * piece; /* lanes past the super-row's end re-read its last pair: */ \
        q_ = q_ < Q.sl - 2 ? q_ : Q.sl - 2;             /* the lines behind it belong to the next super-row      */ \
        const int64_t ibase_ = ((tt) * 64 + lr) * Q.sl + q_;                                              \
        SW_REP16(SW_LOAD)                                                                                 \
    }
#define SW_STAGE(r) *reinterpret_cast<double2 *>(&tile[(SW_RPI * r + lr) * SW_PITCH + 2 * piece]) = v##r;

// EXP: timing experiments only (wrong results): bit 0 = W operands are constants (no scalar loads), bit 1 = g is a constant
// (no LDS reads), bit 2 = no LDS staging writes
template <int C, int EXP = 0>
__global__ __launch_bounds__(SW_THREADS, SW_MINWAVES) void k_ols_sweep(
    const double *__restrict__ G, const double *__restrict__ W, const double *__restrict__ syy,
    const double *__restrict__ tcoef, double *__restrict__ beta, double *__restrict__ var,
    double *__restrict__ pval, const SweepDims D, const SweepGeom Q) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *tile = lds + wave * SW_TILE;
    const double *row = tile + lane * SW_PITCH;
    const int lr = lane / SW_LPR, piece = lane % SW_LPR;
    const int64_t wstride = (int64_t)gridDim.x * SW_WAVES;
    const int64_t lastpair = Q.total - 2;
    const int n = D.n, ld = (int)D.ld;

    SW_REP16(SW_DECL)
    int64_t t = (int64_t)blockIdx.x * SW_WAVES + wave;
    if (t >= D.ntiles) return;
    if (Q.prefetch & 1) SW_ISSUE(t, 0)
    for (; t < D.ntiles; t += wstride) {
        double acc[C];
        double s2 = 0.0, shift = 0.0;
        double stash_b = 0.0, stash_v = 0.0, stash_p = 0.0; // g == 2, one trait: the results of the super-row's first locus wait for
                          // the second, so that the pair leaves as one 16-byte store per lane (two half-filled lines per store
                          // otherwise: WRITE_SIZE 2x)
        int j = 0, pos = 0; // locus within the super-row, position within its row (pools n..ld-1 are padding)
        for (int ch = 0; ch < Q.nch; ++ch) {
            if (!(Q.prefetch & 1)) SW_ISSUE(t, ch)
            if (EXP & 4) {
#define SW_TOUCH(r) s2 += v##r.x + v##r.y;
                SW_REP16(SW_TOUCH)
            } else {
                SW_REP16(SW_STAGE)
            }
            __builtin_amdgcn_wave_barrier();
            if ((Q.prefetch & 3) == 1) {   // next chunk of this tile, else the first chunk of the wave's next tile (clamped loads: harmless past the end)
                const bool more = ch + 1 < Q.nch;
                const int64_t tn = more ? t : t + wstride;
                const int cn = more ? ch + 1 : 0;
                SW_ISSUE(tn, cn)
            }
            const int64_t left = Q.sl - (int64_t)ch * SW_CH;
            const int iend = left < SW_CH ? (int)left : SW_CH;
            int i = (Q.prefetch & 4) ? iend : 0; // (timing experiments, POOLGEN_SWEEP_MODE: bit 1 = no loads after the first, bit 2 = no arithmetic)
            while (i < iend) {
                if (pos >= n) { // padding between n and ld
                    int skip = ld - pos;
                    skip = skip < iend - i ? skip : iend - i;
                    i += skip; pos += skip;
                    if (pos == ld) { pos = 0; ++j; }
                    continue;
                }
                if (pos == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = 0.0;
                    s2 = 0.0;
                    shift = row[i]; // any per-locus constant cancels because Z contains the intercept
                }
                int len = n - pos;
                len = len < iend - i ? len : iend - i;
                const double *Wp = W + (size_t)pos * C;
                if (i == 0 && len == SW_CH) {
#pragma unroll
                    for (int q = 0; q < SW_CH; q += 2) {
                        const double2 g2 = (EXP & 2) ? double2{shift + 1e-3 * q, shift - 1e-3} : *reinterpret_cast<const double2 *>(&row[q]);
                        const double ga = g2.x - shift;
                        const double gb = g2.y - shift;
                        s2 = fma(ga, ga, s2);
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] = fma(ga, (EXP & 1) ? 0.25 + c : Wp[q * C + c], acc[c]);
                        s2 = fma(gb, gb, s2);
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] = fma(gb, (EXP & 1) ? 0.5 - c : Wp[(q + 1) * C + c], acc[c]);
                    }
                } else if (((i | len) & 1) == 0) {
                    // a run that is not a whole chunk (the chunk holds the end of one locus and the start of the next): blocks of
                    // 8 pools with their W operands fetched up front (one or two wide scalar loads per block instead of a
                    // wait per pair), then what is left pair by pair
                    int q = 0;
                    for (; C <= 4 && q + 8 <= len; q += 8) { // (more columns than that and the block's W operands overflow the scalar registers)
                        double2 g2[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) g2[u] = *reinterpret_cast<const double2 *>(&row[i + q + 2 * u]);
                        const double *wq = Wp + (size_t)q * C;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double ga = g2[u].x - shift;
                            const double gb = g2[u].y - shift;
                            s2 = fma(ga, ga, s2);
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[c] = fma(ga, wq[(2 * u) * C + c], acc[c]);
                            s2 = fma(gb, gb, s2);
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[c] = fma(gb, wq[(2 * u + 1) * C + c], acc[c]);
                        }
                    }
                    for (; q < len; q += 2) {
                        const double2 g2 = *reinterpret_cast<const double2 *>(&row[i + q]);
                        const double ga = g2.x - shift;
                        const double gb = g2.y - shift;
                        s2 = fma(ga, ga, s2);
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] = fma(ga, Wp[q * C + c], acc[c]);
                        s2 = fma(gb, gb, s2);
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] = fma(gb, Wp[(q + 1) * C + c], acc[c]);
                    }
                } else { // odd pool counts: one value at a time
                    for (int q = 0; q < len; ++q) {
                        const double ga = row[i + q] - shift;
                        s2 = fma(ga, ga, s2);
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] = fma(ga, Wp[q * C + c], acc[c]);
                    }
                }
                i += len; pos += len;
                if (pos == n) {
                    // ---- per-locus closing arithmetic (gwas/ols.rs:102-116, 139-158) ---------------------
                    const int64_t l = (t * 64 + lane) * Q.g + j;
                    const bool pairs = Q.g == 2 && D.k == 1;
                    if (l < D.p) {
                        double uu = 0.0;
#pragma unroll
                        for (int a = 0; a < C; ++a) uu = (a < D.m1) ? fma(acc[a], acc[a], uu) : uu;
                        const double sgg = s2 - uu;
                        const bool bad = !(sgg > D.tau * s2);
                        for (int jt = 0; jt < D.k; ++jt) {
                            double sgy = 0.0;
#pragma unroll
                            for (int a = 0; a < C; ++a) sgy = (a == D.m1 + jt) ? acc[a] : sgy;
                            double b, vb, pv;
                            ols_close(sgg, sgy, syy[jt], bad, D.dfe, D.tdf, tcoef, D.ntcoef, b, vb, pv);
                            if (!pairs || (j == 0 && l + 1 >= D.p)) { // (or the matrix ends inside this super-row)
                                beta[l * D.k + jt] = b;
                                var[l * D.k + jt] = vb;
                                pval[l * D.k + jt] = pv;
                            } else if (j == 0) {
                                stash_b = b; stash_v = vb; stash_p = pv;
                            } else {
                                *reinterpret_cast<double2 *>(&beta[l - 1]) = double2{stash_b, b};
                                *reinterpret_cast<double2 *>(&var[l - 1]) = double2{stash_v, vb};
                                *reinterpret_cast<double2 *>(&pval[l - 1]) = double2{stash_p, pv};
                            }
                        }
                    }
                    if (n == ld) { pos = 0; ++j; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- the sweep on a per-ROW chunk grid (round 1's form: no super-rows, no prefetch; rows that start 64 bytes into a line make
// it touch 1.14x the algorithmic bytes).  It stays in the product for WIDE designs: measured on 200 pools x 10 M loci with m = 8
// covariates (12 columns) 3.15 ms against 3.6-3.9 ms for the super-row kernel above (whose staging registers, prefetch and
// mid-chunk locus boundaries cost more than the over-fetch once 14 fp64 operations per pool keep the vector unit busy), while
// up to 8 columns (m = 0, 2, 4, 6 with one trait: 2.70 / 2.88 / 2.93 / 2.93 ms against 2.72 / 2.95 / 2.95 / 2.97) the super-row
// kernel is ahead and moves exactly the algorithmic bytes.
// launch_sweep picks by column count; POOLGEN_SWEEP_V1=1 / POOLGEN_SWEEP_V2=1 force one of them (A/B timing).
template <int C>
__global__ __launch_bounds__(SW_THREADS, SW_MINWAVES) void k_ols_sweep_rows(
    const double *__restrict__ G, const double *__restrict__ W, const double *__restrict__ syy,
    const double *__restrict__ tcoef, double *__restrict__ beta, double *__restrict__ var,
    double *__restrict__ pval, const SweepDims D) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *tile = lds + wave * SW_TILE;
    const int nfull = D.n / SW_CH;
    const int ntail = D.n - nfull * SW_CH;
    const int64_t wstride = (int64_t)gridDim.x * SW_WAVES;

    for (int64_t t = (int64_t)blockIdx.x * SW_WAVES + wave; t < D.ntiles; t += wstride) {
        const int64_t l0 = t * 64;
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        double s2 = 0.0, shift = 0.0;
        for (int ch = 0; ch < nfull; ++ch)
            sweep_chunk<C, true>(G, W + (size_t)ch * SW_CH * C, tile, l0, D.p, D.ld, ch * SW_CH,
                                 SW_CH, lane, ch == 0, shift, s2, acc);
        if (ntail)
            sweep_chunk<C, false>(G, W + (size_t)nfull * SW_CH * C, tile, l0, D.p, D.ld,
                                  nfull * SW_CH, ntail, lane, nfull == 0, shift, s2, acc);

        // ---- per-locus closing arithmetic (gwas/ols.rs:102-116, 139-158) ---------------------
        const int64_t l = l0 + lane;
        if (l < D.p) {
            double uu = 0.0;
#pragma unroll
            for (int a = 0; a < C; ++a) uu = (a < D.m1) ? fma(acc[a], acc[a], uu) : uu;
            const double sgg = s2 - uu;
            const bool bad = !(sgg > D.tau * s2);
            for (int j = 0; j < D.k; ++j) {
                double sgy = 0.0;
#pragma unroll
                for (int a = 0; a < C; ++a) sgy = (a == D.m1 + j) ? acc[a] : sgy;
                double b, vb, pv;
                ols_close(sgg, sgy, syy[j], bad, D.dfe, D.tdf, tcoef, D.ntcoef, b, vb, pv);
                beta[l * D.k + j] = b;
                var[l * D.k + j] = vb;
                pval[l * D.k + j] = pv;
            }
        }
    }
}

// ---- the sweep on the matrix cores ------------------------------------------------------------------------------------
// W'g for 16 loci x 16 columns is one v_mfma_f64_16x16x4_f64 chain: A = 16 loci x 4 pools of G, B = 4 pools x 16 columns of
// W = [Q | ytilde], D = 16 loci x 16 columns.  The A operand wants lane (i = lane & 15, kk = lane >> 4) to hold pool kk of
// locus i -- and a lane that loads the 16 bytes at  row i, pools 8 c + 2 kk, 8 c + 2 kk + 1  straight from global memory has
// exactly the A operands of two MFMAs (the pool order inside a chain is free as long as B follows it).  So there is NO LDS
// transposition: G goes HBM -> registers -> matrix pipe, 16 rows x 64 bytes per load instruction, the other half of every
// 128-byte line by the next instruction of the same wave; measured with the arithmetic in place at 6.45 TB/s on 200 pools x
// 10 M loci (tools/mb_msweep.hip), against 6.3 TB/s for the best staged pattern with no arithmetic at all (tools/mb_pattern).
// The vector ALU is left with g - g[0] and g'g (3 operations per element whatever the number of columns), the matrix pipe is
// about a third busy for up to 16 columns, so the kernel is bound by the read of G for every design the product meets.
//   B lives in LDS ([chunk][lane] as double2 = the two pools of the lane: one conflict-free ds_read_b128 per chunk), filled
//   once per workgroup from W; column groups beyond 16 (NCG = 2: up to 32 columns) reuse A with a second accumulator.
//   Loads run U chunks (U x 1 KB per wave) at a time through a ring of R register buffers, so a wave has (R - 1) U .. R U KB
//   in flight at any time, across tile and 64-locus-group boundaries.
//   g'g: per-lane partial sums over the lane's pools, summed over the four kk lanes of a locus by one more MFMA against a
//   matrix of ones -- which also delivers it in the D layout of the other sums.
//   Closing: the D fragments of four 16-locus tiles go through a wave-private LDS stage (64 loci x (columns + 1)), then one
//   lane per locus runs the same closing arithmetic as the other sweep kernels and stores 64 consecutive results.
typedef double ms_d4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
struct MsGeom {
    int nc;         // 8-pool chunks per locus = ceil(n / 8)
    int ng;         // groups of U chunks = ceil(nc / U)
    int cols;       // row pitch of W
    int cu;         // columns in use = m + 1 + k
    int pitch;      // doubles per locus in the closing stage (odd: conflict-free lane-per-locus reads)
    int exp;        // timing experiments (POOLGEN_SWEEP_EXP, wrong results): 1 = no closing arithmetic, 2 = no closing at all,
                    // 8 = no stores, 16 = stores that stay in the L2
    int mask_last;  // ... or past its n pools: those elements are zeroed
    int64_t n64;    // 64-locus groups
};
struct MsCursor { int64_t t64; int T, g; };
// one workgroup of 8 waves per CU shares the B table (4 waves with their 512 registers each for 33 .. 48 columns)
constexpr int ms_threads(int ncg) { return ncg == 3 ? 256 : 512; }

// The closing code's STORES are written in assembly, for the sake of the loads: the compiler's wait-count pass, on seeing a
// loop with vector-memory stores (the loop over traits), puts s_waitcnt vmcnt(0) in front of it -- which drains the ring of
// loads below every 64 loci -- and, merging that path into the head of the main loop, makes the consumer of ring buffer 0
// wait for all outstanding loads, so that one group instead of R is in flight (measured: 3.06 ms on 200 pools x 10 M loci
// against 2.5 ms for the same loop in tools/mb_msweep.hip).  Stores the pass cannot see do not trigger any of that, and they
// can only make its waits for loads longer, never too short: loads return in order, so "at most N operations outstanding"
// implies that a load with N younger loads behind it has landed, whatever stores are also in flight.
// (Writing the LOADS in assembly with hand-placed waits was tried first and is wrong: the compiler is free to copy a
// register it believes defined -- and did, ahead of the wait -- while the load is still in flight.)
__device__ __forceinline__ void ms_store8(double *addr, double x) {
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(addr), "v"(x) : "memory");
}
template <int N, typename F>
__device__ __forceinline__ void ms_static_for(F &&f) {
    if constexpr (N > 0) {
        ms_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// MODE 0: the regression sweep.  MODE 1: the products alone (gp::ols slopes, pg_gp_beta_cols): out = G Z for D.k columns, row-
// or column-major, no shift, optionally the rows' sums of squares in D.ss; `beta` is the output, the other pointers unused.
// MODE 2: the intercept-only sweep (m = 0) that also leaves, per wave, sum_l (sum_i g_li)^2 and sum_l sum_i g_li^2 = 1'S1 and
// trace(S) of the kinship sums S = sum_l g_l g_l' it never forms (the lazy-kinship route of pg_ols_kinship_dev).
template <int U, int R, int NCG, int MODE>
__global__ __launch_bounds__(ms_threads(NCG), 1) void k_ols_sweep_mfma(
    const double *__restrict__ G, const double *__restrict__ W, const double *__restrict__ syy,
    const double *__restrict__ tcoef, double *__restrict__ beta, double *__restrict__ var,
    double *__restrict__ pval, const SweepDims D, const MsGeom M) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (scalar: the cursors below live in SGPRs)
    const int li = lane & 15, lk = lane >> 4;
    const int ncp = M.ng * U;
    double2 *Wl = reinterpret_cast<double2 *>(lds); // [NCG][ncp][64]
    double *stage = lds + (size_t)NCG * ncp * 128 + (size_t)wave * 64 * M.pitch;
    for (int x = threadIdx.x; x < NCG * ncp * 64; x += ms_threads(NCG)) {
        const int ln = x & 63, c = (x >> 6) % ncp, cg = (x >> 6) / ncp;
        const int j = 16 * cg + (ln & 15), pl = 8 * c + 2 * (ln >> 4);
        double2 w = {0.0, 0.0};
        if (j < M.cu) {
            if (pl < D.n) w.x = W[(size_t)pl * M.cols + j];
            if (pl + 1 < D.n) w.y = W[(size_t)(pl + 1) * M.cols + j];
        }
        Wl[x] = w;
    }
    __syncthreads();
    const int64_t wstride = (int64_t)gridDim.x * (ms_threads(NCG) / 64);
    MsCursor ci = {(int64_t)blockIdx.x * (ms_threads(NCG) / 64) + wave, 0, 0};
    if (ci.t64 >= M.n64) return;
    MsCursor cc = ci;
    // A lane's byte offset inside a 16-locus tile never changes; the tile moves through the (scalar) buffer descriptor, whose
    // range check also answers reads past the end of the matrix with zeros: no address arithmetic and no clamps per load.
    const uint32_t lane_off = (uint32_t)((int64_t)li * D.ld * 8 + 16 * lk);

    auto advance = [&](MsCursor &c) {
        if (++c.g < M.ng) return;
        c.g = 0;
        if (++c.T < 4) return;
        c.T = 0;
        c.t64 += wstride;
    };
    auto issue = [&](uint4_t (&v)[U], const MsCursor &c) {
        const int64_t row0 = c.t64 * 64 + 16 * c.T;
        int64_t left = (D.p - row0) * D.ld * 8;
        left = left < 0 ? 0 : (left > 0xFFFFF000ll ? 0xFFFFF000ll : left);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(G + (left ? row0 : 0) * D.ld), 0,
                                                                            (int)(uint32_t)left, 0x00020000);
        const uint32_t vo = lane_off + (uint32_t)(64 * U) * (uint32_t)c.g;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 64u * u, 0, 0);
    };
    ms_d4 acc[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) acc[cg] = ms_d4{0.0, 0.0, 0.0, 0.0};
    double s2 = 0.0, shift = 0.0;
    double lz1 = 0.0, lz2 = 0.0; // MODE 2
    const double lz_invq = (MODE == 2) ? 1.0 / W[0] : 0.0;
    auto as_f64 = [](uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); };

    auto products = [&](uint4_t (&v)[U], const double2 (&bq)[NCG][U], const MsCursor &c, auto masked) {
        ms_static_for<U>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            double ga = as_f64(v[u].x, v[u].y), gb = as_f64(v[u].z, v[u].w);
            if constexpr (MODE != 1) { ga -= shift; gb -= shift; }
            if (decltype(masked)::value) { // the group that holds the row's end: pools n .. are not this locus's
                const int pl = 8 * (U * c.g + u) + 2 * lk;
                ga = pl < D.n ? ga : 0.0;
                gb = pl + 1 < D.n ? gb : 0.0;
            }
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                acc[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, bq[cg][u].x, acc[cg], 0, 0, 0);
                acc[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(gb, bq[cg][u].y, acc[cg], 0, 0, 0);
            }
            s2 = fma(ga, ga, s2);
            s2 = fma(gb, gb, s2);
        });
    };
    auto consume = [&](uint4_t (&v)[U], const MsCursor &c) {
        // the group's B operands first: their LDS latency passes while the wave waits for (or subtracts from) the G values
        double2 bq[NCG][U];
        {
            const double2 *wl = Wl + (size_t)(U * c.g) * 64 + lane;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
                for (int u = 0; u < U; ++u) bq[cg][u] = wl[(size_t)(cg * ncp + u) * 64];
        }
        if (MODE != 1 && c.g == 0) shift = __shfl(as_f64(v[0].x, v[0].y), li); // any per-locus constant cancels because Z contains the intercept
        if constexpr (MODE == 2) { // the closing lane of this locus wants it back (no load there: it would wait for the whole ring)
            if (c.g == 0 && lk == 0) stage[(16 * c.T + li) * M.pitch + M.cu + 1] = shift;
        }
        if (c.g != M.ng - 1) {
            products(v, bq, c, std::false_type{});
            return;
        }
        if (M.mask_last) products(v, bq, c, std::true_type{});
        else products(v, bq, c, std::false_type{});
        // ---- end of a 16-locus tile: fragments -> stage ------------------------------------------------------------
        const ms_d4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(s2, 1.0, ms_d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *sr = stage + (16 * c.T + lk + 4 * r) * M.pitch;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg)
                if (16 * cg + li < M.cu) sr[16 * cg + li] = acc[cg][r];
            if (li == 0) sr[M.cu] = e[r];
        }
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) acc[cg] = ms_d4{0.0, 0.0, 0.0, 0.0};
        s2 = 0.0;
        if (c.T != 3 || (M.exp & 2)) return;
        // ---- end of a 64-locus group: one lane per locus (gwas/ols.rs:102-116, 139-158) --------------------------------
        __builtin_amdgcn_wave_barrier();
        const int64_t l = c.t64 * 64 + lane;
        if (MODE == 1) {
            if (l < D.p) {
                const double *sr = stage + lane * M.pitch;
                for (int a = 0; a < D.k; ++a) ms_store8(D.colmajor ? &beta[(int64_t)a * D.p + l] : &beta[l * D.k + a], sr[a]);
                if (D.ss) ms_store8(&D.ss[l], sr[M.cu]);
            }
        } else if (l < D.p) {
            const double *sr = stage + lane * M.pitch;
            double uu = 0.0;
            for (int a = 0; a < D.m1; ++a) uu = fma(sr[a], sr[a], uu);
            const double gg = sr[M.cu];
            const double sgg = gg - uu;
            const bool bad = !(sgg > D.tau * gg);
            if constexpr (MODE == 2) {
                // g = g' + g0 with g0 the locus' first value: sum g = sum g' + n g0, sum g^2 = sum g'^2 + 2 g0 sum g' + n g0^2, and
                // sum g' = u / q with u = sr[0] the product with the constant column q = W[0][0] = +-1 / sqrt(n) of the basis
                const double g0 = sr[M.cu + 1], s1 = sr[0] * lz_invq, nn = (double)D.n;
                const double sg = fma(nn, g0, s1);
                lz1 = fma(sg, sg, lz1);
                lz2 += fma(nn * g0, g0, fma(2.0 * g0, s1, gg));
            }
            if (D.k == 1) { // straight-line code and ordinary stores
                double b = sgg, vb = gg, pv = uu;
                if (!(M.exp & 1)) ols_close(sgg, sr[D.m1], syy[0], bad, D.dfe, D.tdf, tcoef, D.ntcoef, b, vb, pv);
                beta[l] = b;
                var[l] = vb;
                pval[l] = pv;
            } else {
                for (int jt = 0; jt < D.k; ++jt) {
                    double b, vb, pv;
                    ols_close(sgg, sr[D.m1 + jt], syy[jt], bad, D.dfe, D.tdf, tcoef, D.ntcoef, b, vb, pv);
                    ms_store8(&beta[l * D.k + jt], b);
                    ms_store8(&var[l * D.k + jt], vb);
                    ms_store8(&pval[l * D.k + jt], pv);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // the ring: buffer r is consumed and refilled (for the item R places on) at one place in the loop body
    uint4_t v[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r) { issue(v[r], ci); advance(ci); }
    // A whole number of turns of the ring and no exits from inside it (with a break after every consume() the wait-count pass
    // loses track of the order of the loads and waits for far too many): the items past the wave's last one are loads the
    // descriptor answers with zeros and fits of loci >= p, which are not stored.
    const int64_t cnt64 = (M.n64 - cc.t64 + wstride - 1) / wstride;
    const int64_t turns = (cnt64 * 4 * M.ng + R - 1) / R;
    for (int64_t it = 0; it < turns; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            consume(v[r], cc);
            advance(cc);
            issue(v[r], ci);
            advance(ci);
        }
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { lz1 += __shfl_xor(lz1, off); lz2 += __shfl_xor(lz2, off); }
        if (lane == 0) {
            double *o = D.lz + 2 * ((int64_t)blockIdx.x * (ms_threads(NCG) / 64) + wave);
            o[0] = lz1;
            o[1] = lz2;
        }
    }
}

// gp::ols coefficient pass (gp/ols.rs:47-72): beta_l = sum_i G[l][i] * Z[i][j], the same
// lane-per-locus streaming pass without the regression epilogue (Z = rows of pinv(X X^T) y
// scattered to the training pools, zero elsewhere).
template <int C>
__global__ __launch_bounds__(SW_THREADS) void k_gp_beta(const double *__restrict__ G,
                                                        const double *__restrict__ W,
                                                        double *__restrict__ out, const SweepDims D) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *tile = lds + wave * SW_TILE;
    const int nfull = D.n / SW_CH;
    const int ntail = D.n - nfull * SW_CH;
    const int64_t wstride = (int64_t)gridDim.x * SW_WAVES;
    for (int64_t t = (int64_t)blockIdx.x * SW_WAVES + wave; t < D.ntiles; t += wstride) {
        const int64_t l0 = t * 64;
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        double s2 = 0.0, shift = 0.0;
        for (int ch = 0; ch < nfull; ++ch)
            sweep_chunk<C, true, false>(G, W + (size_t)ch * SW_CH * C, tile, l0, D.p, D.ld, ch * SW_CH, SW_CH,
                                        lane, false, shift, s2, acc);
        if (ntail)
            sweep_chunk<C, false, false>(G, W + (size_t)nfull * SW_CH * C, tile, l0, D.p, D.ld, nfull * SW_CH,
                                         ntail, lane, false, shift, s2, acc);
        const int64_t l = l0 + lane;
        if (l < D.p) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (c < D.k) out[D.colmajor ? (int64_t)c * D.p + l : l * D.k + c] = acc[c];
            if (D.ss) D.ss[l] = s2;
        }
    }
}

// The same pass for many coefficient columns and many pools (the folds of a CV repetition at n = 500: Z is 48 KB).
// There the wave-uniform operands no longer fit the scalar data cache and every s_load of Z went to L2, serialising
// the FMAs behind it (k_gp_beta<12>: 3.2 TB/s).  Here Z sits in LDS for the life of the block (uniform-address
// ds_read_b128 = broadcast) and the global loads of chunk c + 1 are in flight while chunk c is consumed, so one wave per
// SIMD is enough to keep HBM busy.
template <bool FULL>
__device__ __forceinline__ void gpb_load(const double *__restrict__ G, int64_t l0, int64_t p, int64_t ld, int pool0,
                                         int npool, int lane, double2 (&v)[SW_NLD]) {
    const int lr = lane / SW_LPR, piece = lane % SW_LPR;
    int cofs = 2 * piece;
    if (!FULL) {
        const int last = (npool - 1) & ~1;
        cofs = cofs < last ? cofs : last;
    }
#pragma unroll
    for (int r = 0; r < SW_NLD; ++r) {
        int64_t l = l0 + SW_RPI * r + lr;
        l = l < p ? l : p - 1;
        v[r] = *reinterpret_cast<const double2 *>(G + l * ld + pool0 + cofs);
    }
}
template <bool FULL>
__device__ __forceinline__ void gpb_store(double *tile, int npool, int lane, const double2 (&v)[SW_NLD]) {
    const int lr = lane / SW_LPR, piece = lane % SW_LPR;
    const bool col_ok = FULL || 2 * piece < npool, two = FULL || 2 * piece + 1 < npool;
#pragma unroll
    for (int r = 0; r < SW_NLD; ++r) {
        double2 x = v[r];
        x.x = col_ok ? x.x : 0.0;
        x.y = two ? x.y : 0.0;
        *reinterpret_cast<double2 *>(&tile[(SW_RPI * r + lr) * SW_PITCH + 2 * piece]) = x;
    }
}

template <int C>
__global__ __launch_bounds__(SW_THREADS) void k_gp_beta_lds(const double *__restrict__ G, const double *__restrict__ W,
                                                            double *__restrict__ out, const SweepDims D, int wdoubles) {
    static_assert(C % 2 == 0, "even column count");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *Ws = lds;
    double *tile = lds + wdoubles + wave * SW_TILE;
    for (int i = threadIdx.x; i < wdoubles; i += SW_THREADS) Ws[i] = W[i];
    __syncthreads();
    const int nfull = D.n / SW_CH;
    const int ntail = D.n - nfull * SW_CH;
    const int nch = nfull + (ntail ? 1 : 0);
    const int64_t wstride = (int64_t)gridDim.x * SW_WAVES;
    const double *row = tile + lane * SW_PITCH;
    for (int64_t t = (int64_t)blockIdx.x * SW_WAVES + wave; t < D.ntiles; t += wstride) {
        const int64_t l0 = t * 64;
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        double2 v[SW_NLD];
        if (nfull) gpb_load<true>(G, l0, D.p, D.ld, 0, SW_CH, lane, v);
        else gpb_load<false>(G, l0, D.p, D.ld, 0, ntail, lane, v);
        for (int ch = 0; ch < nch; ++ch) {
            const bool full = ch < nfull;
            if (full) gpb_store<true>(tile, SW_CH, lane, v);
            else gpb_store<false>(tile, ntail, lane, v);
            __builtin_amdgcn_wave_barrier();
            if (ch + 1 < nch) { // next chunk's rows travel while this one is consumed
                if (ch + 1 < nfull) gpb_load<true>(G, l0, D.p, D.ld, (ch + 1) * SW_CH, SW_CH, lane, v);
                else gpb_load<false>(G, l0, D.p, D.ld, (ch + 1) * SW_CH, ntail, lane, v);
            }
            const double *wp = Ws + (size_t)ch * SW_CH * C;
            const int np = full ? SW_CH : ((ntail + 1) & ~1); // the tail's odd column was stored as zero
            if (full) {
#pragma unroll
                for (int i = 0; i < SW_CH; i += 2) {
                    const double2 g2 = *reinterpret_cast<const double2 *>(&row[i]);
#pragma unroll
                    for (int c = 0; c < C; c += 2) {
                        const double2 wa = *reinterpret_cast<const double2 *>(&wp[i * C + c]);
                        acc[c] = fma(g2.x, wa.x, acc[c]);
                        acc[c + 1] = fma(g2.x, wa.y, acc[c + 1]);
                    }
#pragma unroll
                    for (int c = 0; c < C; c += 2) {
                        const double2 wb = *reinterpret_cast<const double2 *>(&wp[(i + 1) * C + c]);
                        acc[c] = fma(g2.y, wb.x, acc[c]);
                        acc[c + 1] = fma(g2.y, wb.y, acc[c + 1]);
                    }
                }
            } else {
                for (int i = 0; i < np; i += 2) {
                    const double2 g2 = *reinterpret_cast<const double2 *>(&row[i]);
#pragma unroll
                    for (int c = 0; c < C; c += 2) {
                        const double2 wa = *reinterpret_cast<const double2 *>(&wp[i * C + c]);
                        acc[c] = fma(g2.x, wa.x, acc[c]);
                        acc[c + 1] = fma(g2.x, wa.y, acc[c + 1]);
                    }
#pragma unroll
                    for (int c = 0; c < C; c += 2) {
                        const double2 wb = *reinterpret_cast<const double2 *>(&wp[(i + 1) * C + c]);
                        acc[c] = fma(g2.y, wb.x, acc[c]);
                        acc[c + 1] = fma(g2.y, wb.y, acc[c + 1]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        const int64_t l = l0 + lane;
        if (l < D.p) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (c < D.k) out[D.colmajor ? (int64_t)c * D.p + l : l * D.k + c] = acc[c];
        }
    }
}

// The coefficient pass as a skinny fp64 MFMA GEMM: out (64 loci x 16 columns per wave tile) = G-tile (64 x n) * Z (n x 16).
// v_mfma_f64_16x16x4_f64 with A = 16 loci x 4 pools from the staged tile, B = 4 pools x 16 columns of Z: one 512-byte
// conflict-free LDS read of Z per k-step serves four locus groups, where the VALU form needs a broadcast read per FMA
// operand (k_gp_beta_lds is bound by that LDS return traffic).  Z columns beyond the fits are zero; pools beyond n are
// zero rows of Z and zero columns of the tile, so every chunk runs the same 8 k-steps.  Output column-major (k x p).
constexpr int MB_PITCH = SW_CH + 4;          // doubles per tile row: 8 * 36 bytes puts the 16 loci of an A fragment on all banks
constexpr int MB_TILE = 64 * MB_PITCH;
typedef double mb_double4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(SW_THREADS) void k_gp_beta_mfma(const double *__restrict__ G, const double *__restrict__ Z16,
                                                             double *__restrict__ out, const SweepDims D, int zrows) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int fi = lane & 15, kq = lane >> 4;
    double *Zs = lds;                                  // zrows x 16
    double *tile = lds + (size_t)zrows * 16 + wave * MB_TILE;
    for (int i = threadIdx.x; i < zrows * 16; i += SW_THREADS) Zs[i] = Z16[i];
    __syncthreads();
    const int nch = (D.n + SW_CH - 1) / SW_CH;
    const int nfull = D.n / SW_CH;
    const int ntail = D.n - nfull * SW_CH;
    const int64_t wstride = (int64_t)gridDim.x * SW_WAVES;
    const int lr = lane / SW_LPR, piece = lane % SW_LPR;
    for (int64_t t = (int64_t)blockIdx.x * SW_WAVES + wave; t < D.ntiles; t += wstride) {
        const int64_t l0 = t * 64;
        mb_double4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mb_double4{0.0, 0.0, 0.0, 0.0};
        double2 v[SW_NLD];
        if (nfull) gpb_load<true>(G, l0, D.p, D.ld, 0, SW_CH, lane, v);
        else gpb_load<false>(G, l0, D.p, D.ld, 0, ntail, lane, v);
        for (int ch = 0; ch < nch; ++ch) {
            const bool full = ch < nfull;
            const bool col_ok = full || 2 * piece < ntail, two = full || 2 * piece + 1 < ntail;
#pragma unroll
            for (int r = 0; r < SW_NLD; ++r) {
                double2 x = v[r];
                x.x = col_ok ? x.x : 0.0;
                x.y = two ? x.y : 0.0;
                *reinterpret_cast<double2 *>(&tile[(SW_RPI * r + lr) * MB_PITCH + 2 * piece]) = x;
            }
            __builtin_amdgcn_wave_barrier();
            if (ch + 1 < nch) { // next chunk's rows travel while this one is multiplied
                if (ch + 1 < nfull) gpb_load<true>(G, l0, D.p, D.ld, (ch + 1) * SW_CH, SW_CH, lane, v);
                else gpb_load<false>(G, l0, D.p, D.ld, (ch + 1) * SW_CH, ntail, lane, v);
            }
            const double *zp = Zs + (size_t)(ch * SW_CH + kq) * 16 + fi;
            const double *ap = tile + fi * MB_PITCH + kq;
#pragma unroll
            for (int ks = 0; ks < SW_CH / 4; ++ks) {
                const double b = zp[ks * 64];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[g * 16 * MB_PITCH + 4 * ks], b, acc[g], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // D layout: col = lane & 15, row = (lane >> 4) + 4 * reg -> through the (now idle) tile so that a lane stores
        // its own locus, 64 consecutive doubles of a column per instruction
        double *tr = tile; // [16 columns][64 loci]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[fi * 64 + 16 * g + kq + 4 * r] = acc[g][r];
        __builtin_amdgcn_wave_barrier();
        const int64_t l = l0 + lane;
        if (l < D.p) {
            for (int c = 0; c < D.k; ++c) out[(int64_t)c * D.p + l] = tr[c * 64 + lane];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Intercept-only fits closed from the sums the fused kinship pass left behind (pg_set_phenotypes):
// spec[l] = { sum g', sum g'^2, sum g' ytil_t }, g' = g - g[0].  With Z = [1]: u = sum g' / sqrt(n).
__global__ void k_sweep_finish(const double *__restrict__ spec, const double *__restrict__ syy,
                               const double *__restrict__ tcoef, double *__restrict__ beta,
                               double *__restrict__ var, double *__restrict__ pval, const SweepDims D) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= D.p) return;
    const double *sp = spec + l * (2 + D.k);
    const double s1 = sp[0], s2 = sp[1];
    const double sgg = s2 - s1 * s1 / (double)D.n;
    const bool bad = !(sgg > D.tau * s2);
    for (int j = 0; j < D.k; ++j) {
        double b, vb, pv;
        ols_close(sgg, sp[2 + j], syy[j], bad, D.dfe, D.tdf, tcoef, D.ntcoef, b, vb, pv);
        beta[l * D.k + j] = b;
        var[l * D.k + j] = vb;
        pval[l * D.k + j] = pv;
    }
}

// One trait (the headline): two loci per lane, `half` apart, their p-value series behind one stream of coefficient loads
// (pg_t_two_sided_p_x2).  The arithmetic per locus is ols_close's, operation for operation.
__global__ __launch_bounds__(256) void k_sweep_finish_x2(const double *__restrict__ spec, const double *__restrict__ syy,
                                                         const double *__restrict__ tcoef, double *__restrict__ beta,
                                                         double *__restrict__ var, double *__restrict__ pval, const SweepDims D,
                                                         int64_t half) {
    const int64_t la = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, lb = la + half;
    if (la >= half || la >= D.p) return; // (half is rounded up to the workgroup: fewer than 2 x 256 loci leave lanes with nothing)
    const bool hasb = lb < D.p;
    const double *pa = spec + la * 3, *pb = spec + (hasb ? lb : la) * 3;
    double ba, va, pva, bb, vbb, pvb, tta, ttb;
    auto head = [&](const double *sp, double &b, double &vb, double &tt) -> bool { // false: rank-deficient, everything NaN
        const double s1 = sp[0], s2 = sp[1];
        const double sgg = s2 - s1 * s1 / (double)D.n;
        b = NAN; vb = NAN; tt = 0.0;
        if (!(sgg > D.tau * s2)) return false;
        b = sp[2] / sgg;
        double rss = syy[0] - sp[2] * b;
        rss = rss < 0.0 ? 0.0 : rss;
        vb = (rss / D.dfe) / sgg;
        tt = (fabs(b) <= PG_EPS) ? 0.0 : b / sqrt(vb);
        return true;
    };
    const bool oka = head(pa, ba, va, tta), okb = head(pb, bb, vbb, ttb);
    pg_t_two_sided_p_x2(fabs(tta), fabs(ttb), D.tdf, tcoef, D.ntcoef, pva, pvb);
    pva = !oka ? NAN : (fabs(tta) <= PG_EPS || isnan(tta)) ? 1.0 : pva;
    pvb = !okb ? NAN : (fabs(ttb) <= PG_EPS || isnan(ttb)) ? 1.0 : pvb;
    beta[la] = ba; var[la] = va; pval[la] = pva;
    if (hasb) { beta[lb] = bb; var[lb] = vbb; pval[lb] = pvb; }
}

struct SweepArgs {
    const double *G, *W, *syy, *tcoef;
    double *beta, *var, *pval;
    SweepDims D;
    SweepGeom Q;
};

template <int C>
int launch_sweep(pg_ctx *ctx, const SweepArgs &A, int grid) {
    const size_t shmem = (size_t)SW_WAVES * SW_TILE * sizeof(double);
    const bool rows_kernel = std::getenv("POOLGEN_SWEEP_V1") || ((C >= 12 || A.D.n <= 32) && !std::getenv("POOLGEN_SWEEP_V2"));
    if (rows_kernel) {
        SweepDims D1 = A.D;
        D1.ntiles = (A.D.p + 63) / 64;
        int64_t blocks = (D1.ntiles + SW_WAVES - 1) / SW_WAVES;
        const int64_t cap = (int64_t)ctx->cus * 8;
        const int g1 = (int)(blocks < cap ? blocks : cap);
        PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_ols_sweep_rows<C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        pg_prof_begin(ctx, PG_K_SWEEP);
        hipLaunchKernelGGL(k_ols_sweep_rows<C>, dim3(g1), dim3(SW_THREADS), shmem, ctx->stream, A.G, A.W,
                           A.syy, A.tcoef, A.beta, A.var, A.pval, D1);
        pg_prof_end(ctx);
        PG_HIP(ctx, hipGetLastError());
        return PG_OK;
    }
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_ols_sweep<C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if constexpr (C == 2) {
        if (const char *e = std::getenv("POOLGEN_SWEEP_EXP")) {
            const int x = std::atoi(e);
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                pg_prof_begin(ctx, PG_K_SWEEP);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(SW_THREADS), shmem, ctx->stream, A.G, A.W, A.syy, A.tcoef, A.beta, A.var,
                                   A.pval, A.D, A.Q);
                pg_prof_end(ctx);
            };
            switch (x) {
            case 1: go(k_ols_sweep<2, 1>); break;
            case 2: go(k_ols_sweep<2, 2>); break;
            case 3: go(k_ols_sweep<2, 3>); break;
            case 4: go(k_ols_sweep<2, 4>); break;
            case 6: go(k_ols_sweep<2, 6>); break;
            default: go(k_ols_sweep<2, 7>); break;
            }
            PG_HIP(ctx, hipGetLastError());
            return PG_OK;
        }
    }
    pg_prof_begin(ctx, PG_K_SWEEP);
    hipLaunchKernelGGL(k_ols_sweep<C>, dim3(grid), dim3(SW_THREADS), shmem, ctx->stream, A.G, A.W,
                       A.syy, A.tcoef, A.beta, A.var, A.pval, A.D, A.Q);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

// The matrix-core sweep for up to 48 columns.  U (chunks per load group) is the one of 5 .. 8 that pads the ceil(n / 8) chunks
// of a locus least (200 pools: 25 chunks = 5 groups of 5; 100 pools: 13 -> 14 = 2 groups of 7); the ring is 3 deep for U <= 6,
// else 2.  Fewer than 33 pools stay with the row kernel: measured with 20 and 30 pools x 10-20 M loci, 0.50 / 0.56-0.60 of the
// HBM peak here against 0.61 / 0.62 there (rows this short are mostly closing arithmetic per byte, and padding below 5 chunks).
constexpr int MS_MAX_COLS = 48;
template <int U, int R, int NCG, int MODE>
int launch_sweep_mfma_as(pg_ctx *ctx, const SweepArgs &A, MsGeom M, int kernel_id) {
    constexpr int threads = ms_threads(NCG), waves = threads / 64;
    M.ng = (M.nc + U - 1) / U;
    const int ncp = M.ng * U;
    M.exp = std::getenv("POOLGEN_SWEEP_EXP") ? std::atoi(std::getenv("POOLGEN_SWEEP_EXP")) : 0;
    M.mask_last = 8 * ncp > A.D.n;
    const size_t shmem = ((size_t)NCG * ncp * 128 + (size_t)waves * 64 * M.pitch) * sizeof(double);
    auto kern = k_ols_sweep_mfma<U, R, NCG, MODE>;
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    int per_cu = 0;
    PG_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, shmem));
    if (per_cu < 1) return pg_fail(ctx, PG_ERR_UNSUPPORTED, "sweep: %zu bytes of LDS per workgroup do not fit", shmem);
    if (const char *e = std::getenv("POOLGEN_SWEEP_GRID_MULT")) per_cu = std::max(1, std::atoi(e)); // experiments
    const int64_t blocks = (M.n64 + waves - 1) / waves, cap = (int64_t)ctx->cus * per_cu;
    pg_prof_begin(ctx, kernel_id);
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(threads), shmem, ctx->stream, A.G, A.W, A.syy,
                       A.tcoef, A.beta, A.var, A.pval, A.D, M);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

int ms_pick_u(int nc) {
    int U = 8, best = 1 << 30;
    for (int u = 8; u >= 5; --u) {
        const int padded = (nc + u - 1) / u * u;
        if (padded < best) { best = padded; U = u; }
    }
    if (const char *e = std::getenv("POOLGEN_SWEEP_U")) { const int u = std::atoi(e); if (u >= 5 && u <= 8) U = u; } // experiments
    return U;
}
// does the B table ([column groups][padded chunks] KB) plus the closing stage fit the 160 KB of a CU?
bool ms_fits(int n, int cu) {
    const int nc = (n + 7) / 8, U = ms_pick_u(nc), ncp = (nc + U - 1) / U * U, ncg = (cu + 15) / 16;
    return nc >= 5 && cu <= MS_MAX_COLS && ((size_t)ncg * ncp * 128 + (size_t)(ms_threads(ncg) / 64) * 64 * ((cu + 1) | 1)) * sizeof(double) <= 160 * 1024;
}

template <int MODE>
int launch_sweep_mfma(pg_ctx *ctx, const SweepArgs &A, int cols, int cu, int kernel_id) {
    MsGeom M;
    M.nc = (A.D.n + 7) / 8;
    M.cols = cols;
    M.cu = cu;
    M.pitch = (MODE == 2) ? ((cu + 2) | 1) : ((cu + 1) | 1); // MODE 2 keeps the locus' shift behind the sums
    M.n64 = (A.D.p + 63) / 64;
    const int U = ms_pick_u(M.nc);
    const int ncg = (cu + 15) / 16;
#define MS_GO(UU, R1, RR)                                                                                         \
    return ncg == 1 ? launch_sweep_mfma_as<UU, R1, 1, MODE>(ctx, A, M, kernel_id)                                 \
         : ncg == 2 ? launch_sweep_mfma_as<UU, RR, 2, MODE>(ctx, A, M, kernel_id)                                 \
                    : launch_sweep_mfma_as<UU, RR, 3, MODE>(ctx, A, M, kernel_id);
    if (const char *e = std::getenv("POOLGEN_SWEEP_R")) { // experiments: other ring depths (one accumulator, the sweep only)
        const int r = std::atoi(e);
        if constexpr (MODE == 0) {
            if (ncg == 1 && U == 5 && r == 2) return launch_sweep_mfma_as<5, 2, 1, 0>(ctx, A, M, kernel_id);
            if (ncg == 1 && U == 5 && r == 4) return launch_sweep_mfma_as<5, 4, 1, 0>(ctx, A, M, kernel_id);
            if (ncg == 1 && U == 7 && r == 2) return launch_sweep_mfma_as<7, 2, 1, 0>(ctx, A, M, kernel_id);
            if (ncg == 1 && U == 8 && r == 2) return launch_sweep_mfma_as<8, 2, 1, 0>(ctx, A, M, kernel_id);
        }
    }
    if constexpr (MODE == 2) { // intercept + traits only: one accumulator
        switch (U) {
        case 5: return launch_sweep_mfma_as<5, 3, 1, 2>(ctx, A, M, kernel_id);
        case 6: return launch_sweep_mfma_as<6, 3, 1, 2>(ctx, A, M, kernel_id);
        case 7: return launch_sweep_mfma_as<7, 3, 1, 2>(ctx, A, M, kernel_id);
        default: return launch_sweep_mfma_as<8, 3, 1, 2>(ctx, A, M, kernel_id);
        }
    } else
    switch (U) { // ring depth: 3 with one accumulator (205-223 registers at U = 7, 8), 2 for U >= 7 with more (measured: +2 % at n = 500)
    case 5: MS_GO(5, 3, 3)
    case 6: MS_GO(6, 3, 3)
    case 7: MS_GO(7, 3, 2)
    default: MS_GO(8, 3, 2)
    }
#undef MS_GO
}

int round_cols(int c) {
    const int sizes[] = {2, 3, 4, 6, 8, 12, 16, 24, PG_MAX_SWEEP_COLS};
    for (int s : sizes)
        if (c <= s) return s;
    return -1;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// host set-up: basis of [1 | C], projected phenotypes, W upload
// ---------------------------------------------------------------------------------------------
extern "C" int pg_covariates_set(pg_ctx *ctx, int n, const double *Cmat, int m, const double *Y,
                                 int k) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, n >= 2 && k >= 1 && m >= 0 && Y, "covariates: bad shape n=%d m=%d k=%d", n, m, k);
    PG_CHECK(ctx, m == 0 || Cmat, "covariates: C is null with m=%d", m);
    // every route to a fit passes here (pg_kinship_set's fast exits included): n - P residual degrees of freedom (gwas/ols.rs:103)
    if (m + 2 >= n)
        return pg_fail(ctx, PG_ERR_UNSUPPORTED,
                       "n_eigenvecs = %d leaves no residual degrees of freedom with n = %d pools (reference regime n - P <= 0, "
                       "gwas/ols.rs:103); lower --xxt-eigen-variance-explained or add pools", m, n);
    for (int i = 0; i < n * k; ++i)
        PG_CHECK(ctx, !std::isnan(Y[i]), "covariates: phenotype matrix contains NaN; remove pools "
                                           "with missing phenotypes first (gwas/ols.rs:287)");
    // the steady state of a repeated analysis (same pools, same traits, intercept only): everything this
    // call would compute and upload is already on the device
    if (m == 0 && ctx->st_m == 0 && ctx->st_n == n && ctx->st_k == k && ctx->W_dev && ctx->tcoef_dev &&
        ctx->tcoef_df == n - 1 && ctx->st_Y.size() == (size_t)n * k &&
        std::memcmp(ctx->st_Y.data(), Y, sizeof(double) * n * k) == 0) {
        ctx->st_Y_matches_ph = (ctx->ph_n == n && ctx->ph_k == k && ctx->ph_Y.size() == (size_t)n * k &&
                                std::memcmp(ctx->ph_Y.data(), Y, sizeof(double) * n * k) == 0);
        return PG_OK;
    }
    const int m1 = m + 1;
    const int cols = round_cols(m1 + k);
    if (cols < 0)
        return pg_fail(ctx, PG_ERR_UNSUPPORTED,
                       "m + 1 + k = %d exceeds the %d columns one sweep launch carries", m1 + k,
                       PG_MAX_SWEEP_COLS);
    PG_HIP(ctx, hipSetDevice(ctx->device));
    // Z = [1 | C]
    std::vector<double> Z((size_t)n * m1), Q((size_t)n * m1);
    for (int i = 0; i < n; ++i) {
        Z[(size_t)i * m1] = 1.0;
        for (int j = 0; j < m; ++j) Z[(size_t)i * m1 + 1 + j] = Cmat[(size_t)i * m + j];
    }
    const int rank = pg_thin_qr(Z.data(), n, m1, Q.data());
    const int n_even = (n + 1) & ~1;
    std::vector<double> W((size_t)n_even * cols, 0.0), syy(k, 0.0), yt(n);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < rank; ++a) W[(size_t)i * cols + a] = Q[(size_t)i * m1 + a];
    for (int j = 0; j < k; ++j) {
        for (int i = 0; i < n; ++i) yt[i] = Y[(size_t)i * k + j];
        for (int pass = 0; pass < 2; ++pass)
            for (int a = 0; a < rank; ++a) {
                double d = 0.0;
                for (int i = 0; i < n; ++i) d += Q[(size_t)i * m1 + a] * yt[i];
                for (int i = 0; i < n; ++i) yt[i] -= d * Q[(size_t)i * m1 + a];
            }
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            W[(size_t)i * cols + m1 + j] = yt[i];
            s += yt[i] * yt[i];
        }
        syy[j] = s;
    }
    const size_t wbytes = W.size() * sizeof(double);
    if (wbytes > ctx->W_cap) {
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->W_dev) PG_HIP(ctx, hipFree(ctx->W_dev));
        ctx->W_dev = nullptr;
        ctx->W_cap = 0;
        PG_HIP(ctx, hipMalloc((void **)&ctx->W_dev, wbytes));
        ctx->W_cap = wbytes;
    }
    if (!ctx->syy_dev) PG_HIP(ctx, hipMalloc((void **)&ctx->syy_dev, sizeof(double) * 66));
    PG_CHECK(ctx, k <= 64, "covariates: at most 64 traits per call");
    PG_HIP(ctx, hipMemcpyAsync(ctx->W_dev, W.data(), wbytes, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(ctx->syy_dev, syy.data(), sizeof(double) * k, hipMemcpyHostToDevice,
                               ctx->stream));
    const int df = n - 1; // StudentsT::new(0, 1, n - 1), gwas/ols.rs:139
    if (ctx->tcoef_df != df || !ctx->tcoef_dev) {
        std::vector<double> tc = pg_tdist_coef(df);
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->tcoef_dev) PG_HIP(ctx, hipFree(ctx->tcoef_dev));
        ctx->tcoef_dev = nullptr;
        PG_HIP(ctx, hipMalloc((void **)&ctx->tcoef_dev, sizeof(double) * (tc.size() + 1)));
        if (!tc.empty())
            PG_HIP(ctx, hipMemcpyAsync(ctx->tcoef_dev, tc.data(), sizeof(double) * tc.size(),
                                       hipMemcpyHostToDevice, ctx->stream));
        ctx->tcoef_df = df;
        ctx->tcoef_len = (int)tc.size();
    }
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // W/syy are stack-owned vectors
    ctx->st_n = n;
    ctx->st_m = m;
    ctx->st_k = k;
    ctx->st_cols = cols;
    if (m == 0) ctx->st_Y.assign(Y, Y + (size_t)n * k);
    else ctx->st_Y.clear();
    ctx->st_Y_matches_ph = (ctx->ph_n == n && ctx->ph_k == k && ctx->ph_Y.size() == (size_t)n * k &&
                            std::memcmp(ctx->ph_Y.data(), Y, sizeof(double) * n * k) == 0);
    return PG_OK;
}

namespace {
// sum of all entries and trace of S (n x n): what the first fast exit of the n_eigenvecs rule needs, without the matrix
// crossing the bus.  One workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void k_sum_trace(const double *__restrict__ S, int n, double *__restrict__ out) {
    __shared__ double sm[2][1024];
    double tot = 0.0, tr = 0.0;
    for (int i = threadIdx.x; i < n * n; i += 1024) tot += S[i];
    for (int i = threadIdx.x; i < n; i += 1024) tr += S[(size_t)i * n + i];
    sm[0][threadIdx.x] = tot; sm[1][threadIdx.x] = tr;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { sm[0][threadIdx.x] += sm[0][threadIdx.x + off]; sm[1][threadIdx.x] += sm[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sm[0][0]; out[1] = sm[1][0]; }
}
} // namespace

extern "C" int pg_kinship_set(pg_ctx *ctx, const double *S_dev, int64_t p_total, int n,
                              const double *Y, int k, double var_explained, int force_m, int *m_out,
                              double *K_out, double *evals_out) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, S_dev && p_total > 0 && n >= 2, "kinship_set: bad arguments");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> K((size_t)n * n), ev(n), V;
    {
        // through the context's pinned buffer: a D2H into pageable memory is staged by the runtime
        int prc = pg_pin_reserve(ctx, sizeof(double) * ((size_t)n * n + 2));
        if (prc) return prc;
        if (force_m < 0 && !evals_out && !K_out) {
            // the common outcome (m = 0 by the Rayleigh quotient of the ones vector, see below) is decided from two numbers
            // formed on the device; only when that test does not settle it does the matrix come over
            if (!ctx->syy_dev) PG_HIP(ctx, hipMalloc((void **)&ctx->syy_dev, sizeof(double) * 66));
            double *two = ctx->syy_dev + 64; // behind the 64 trait slots
            hipLaunchKernelGGL(k_sum_trace, dim3(1), dim3(1024), 0, ctx->stream, S_dev, n, two);
            PG_HIP(ctx, hipGetLastError());
            double *hp = static_cast<double *>(ctx->pin) + (size_t)n * n;
            PG_HIP(ctx, hipMemcpyAsync(hp, two, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream));
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const double tot = hp[0], tr = hp[1];
            if (tr > 0.0 && std::isfinite(tot) && (tot / n) / tr >= var_explained + 1e-9) {
                if (m_out) *m_out = 0;
                return pg_covariates_set(ctx, n, nullptr, 0, Y, k);
            }
        }
        PG_HIP(ctx, hipMemcpyAsync(ctx->pin, S_dev, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(K.data(), ctx->pin, sizeof(double) * (size_t)n * n);
    }
    for (int i = 0; i < n; ++i) // the diagonal is a sum of squares: NaN there <=> some frequency of pool i is NaN
        if (std::isnan(K[(size_t)i * n + i]))
            return pg_fail(ctx, PG_ERR_INVALID, "kinship_set: the kinship matrix contains NaN -- pool %d has no coverage at some locus "
                           "that passed the filters (its frequencies are NaN, base/sync.rs:176-183); the reference's eig() fails on such a matrix too", i);
    const double pd = (double)p_total;
    auto scale_K = [&]() { for (auto &x : K) x = x / pd; }; // kinship = G G^T / p  (gwas/ols.rs:295)
    int m = force_m;
    if (force_m < 0 && !evals_out) {
        // Fast exits of the n_eigenvecs rule (gwas/ols.rs:297-311): with eigenvalues descending,
        // cum[0] = lambda_1 / sum(lambda) >= threshold already gives m = 0, and sum(lambda) is
        // trace(K).  Both ratios are scale-free, so they are formed on S itself.
        // (1) lambda_1 >= v'Kv for ANY unit v: with v = 1/sqrt(n) that is sum(K)/n, one pass over K,
        //     and for an uncentred kinship it is already within ~1e-3 of lambda_1.
        // (2) lambda_1 by power iteration (lambda_2/lambda_1 ~ 1e-3: a handful of n^2 mat-vecs) instead
        //     of the O(n^3) decomposition.  Anything not clearly above the threshold falls through to
        //     the full solver.
        double tr = 0.0, tot = 0.0;
        for (int i = 0; i < n; ++i) tr += K[(size_t)i * n + i];
        for (size_t i = 0; i < (size_t)n * n; ++i) tot += K[i];
        bool zero = tr > 0.0 && (tot / n) / tr >= var_explained + 1e-9;
        if (!zero) {
            std::vector<double> v(n, 1.0 / std::sqrt((double)n)), w(n);
            double lam = 0.0, prev = -1.0;
            bool conv = false;
            for (int it = 0; it < 60 && !conv; ++it) {
                double nrm = 0.0, rq = 0.0;
                for (int i = 0; i < n; ++i) {
                    double sacc = 0.0;
                    const double *row = &K[(size_t)i * n];
                    for (int j = 0; j < n; ++j) sacc += row[j] * v[j];
                    w[i] = sacc;
                    rq += sacc * v[i];
                    nrm += sacc * sacc;
                }
                nrm = std::sqrt(nrm);
                if (!(nrm > 0.0)) break;
                for (int i = 0; i < n; ++i) v[i] = w[i] / nrm;
                lam = rq;
                conv = std::fabs(lam - prev) <= 1e-14 * std::fabs(lam);
                prev = lam;
            }
            zero = conv && tr > 0.0 && lam / tr >= var_explained + 1e-9;
        }
        if (zero) {
            if (m_out) *m_out = 0;
            if (K_out) { scale_K(); std::memcpy(K_out, K.data(), sizeof(double) * n * n); }
            return pg_covariates_set(ctx, n, nullptr, 0, Y, k);
        }
    }
    scale_K();
    // eigenvalues always; eigenvectors only the m leading ones (pg_sym_eig_top)
    if (force_m > 0) {
        PG_CHECK(ctx, force_m <= n, "kinship_set: force_m=%d exceeds n=%d", force_m, n);
        V.resize((size_t)n * force_m);
        if (pg_sym_eig_top(K.data(), n, force_m, ev.data(), V.data()) != 0)
            return pg_fail(ctx, PG_ERR_INVALID, "kinship_set: eigen-decomposition did not converge");
    } else {
        if (pg_sym_eig(K.data(), n, ev.data(), nullptr, false) != 0)
            return pg_fail(ctx, PG_ERR_INVALID, "kinship_set: eigen-decomposition did not converge");
    }
    if (force_m < 0) {
        // n_eigenvecs rule, literal (gwas/ols.rs:297-311), eigenvalues descending
        double sum = 0.0;
        for (int i = 0; i < n; ++i) sum = sum + ev[i];
        std::vector<double> cum(n);
        for (int i = 0; i < n; ++i) cum[i] = ev[i] / sum;
        m = n;
        for (int i = 1; i < n; ++i) {
            cum[i] = cum[i - 1] + cum[i];
            if ((cum[i - 1] >= var_explained) & (i - 1 < m)) m = i - 1;
        }
        if (m > 0 && m + 2 < n) {
            V.resize((size_t)n * m);
            if (pg_sym_eig_top(K.data(), n, m, ev.data(), V.data()) != 0)
                return pg_fail(ctx, PG_ERR_INVALID, "kinship_set: eigen-decomposition did not converge");
        }
    }
    PG_CHECK(ctx, m <= n, "kinship_set: force_m=%d exceeds n=%d", m, n);
    if (m_out) *m_out = m;
    if (K_out) std::memcpy(K_out, K.data(), sizeof(double) * n * n);
    if (evals_out) std::memcpy(evals_out, ev.data(), sizeof(double) * n);
    if (m + 2 >= n)
        return pg_fail(ctx, PG_ERR_UNSUPPORTED,
                       "n_eigenvecs = %d leaves no residual degrees of freedom with n = %d pools "
                       "(reference regime n - P <= 0, gwas/ols.rs:103); lower "
                       "--xxt-eigen-variance-explained", m, n);
    std::vector<double> C((size_t)n * (m > 0 ? m : 1));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) C[(size_t)i * m + j] = V[(size_t)i * m + j]; // ols.rs:312-315
    return pg_covariates_set(ctx, n, C.data(), m, Y, k);
}

extern "C" int pg_ols_sweep_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld,
                                double *beta_dev, double *var_dev, double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    if (ctx->st_m < 0 || ctx->st_n != n)
        return pg_fail(ctx, PG_ERR_STATE, "sweep: call pg_kinship_set / pg_covariates_set for n=%d first", n);
    PG_CHECK(ctx, G_dev && beta_dev && var_dev && pval_dev, "sweep: null pointer");
    PG_CHECK(ctx, p > 0, "sweep: p must be positive");
    PG_CHECK(ctx, ld >= n && (ld % 2) == 0, "sweep: ld (%lld) must be even and >= n (%d)",
             (long long)ld, n);
    PG_CHECK(ctx, (reinterpret_cast<uintptr_t>(G_dev) & 15) == 0, "sweep: G must be 16-byte aligned");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int cus = ctx->cus;
    SweepArgs P;
    P.G = G_dev; P.W = ctx->W_dev; P.syy = ctx->syy_dev; P.tcoef = ctx->tcoef_dev;
    P.beta = beta_dev; P.var = var_dev; P.pval = pval_dev;
    // super-rows: g consecutive loci whose g * ld doubles start and end on a 128-byte line (see k_ols_sweep)
    {
        int64_t bytes = 8 * ld, gc = 128;
        while (bytes % gc) gc >>= 1; // gcd(8 ld, 128): 8 ld is a multiple of 16
        P.Q.g = (int)(128 / gc);
        P.Q.sl = (int64_t)P.Q.g * ld;
        P.Q.nch = (int)((P.Q.sl + SW_CH - 1) / SW_CH);
        P.Q.total = p * ld;
        P.Q.prefetch = std::getenv("POOLGEN_SWEEP_NOPF") ? 0 : 1;
        if (const char *e = std::getenv("POOLGEN_SWEEP_MODE")) P.Q.prefetch = std::atoi(e); // 3: compute only, 5: memory only (wrong results)
    }
    const int64_t nsr = (p + P.Q.g - 1) / P.Q.g;
    P.D.p = p; P.D.ld = ld; P.D.ntiles = (nsr + 63) / 64;
    P.D.n = n; P.D.m1 = ctx->st_m + 1; P.D.k = ctx->st_k;
    P.D.tdf = ctx->tcoef_df; P.D.ntcoef = ctx->tcoef_len;
    P.D.dfe = (double)n - (double)(ctx->st_m + 2);
    P.D.tau = 1e-12;
    P.D.colmajor = 0;
    P.D.ss = nullptr;
    P.D.lz = nullptr;
    if (ctx->st_m == 0 && ctx->spec_valid && ctx->spec_G == G_dev && ctx->spec_p == p && ctx->spec_n == n &&
        ctx->spec_ld == ld && ctx->spec_k == ctx->st_k && ctx->ph_n == n && ctx->st_Y_matches_ph) {
        // m = 0: the kinship pass already formed the sums of the intercept-only fits from its read of G
        pg_prof_begin(ctx, PG_K_SWEEP_FINISH);
        if (ctx->st_k == 1 && !std::getenv("POOLGEN_FINISH_X1")) {
            const int64_t half = ((p + 1) / 2 + 255) / 256 * 256;
            hipLaunchKernelGGL(k_sweep_finish_x2, dim3((unsigned)(half / 256)), dim3(256), 0, ctx->stream, ctx->spec_dev, ctx->syy_dev,
                               ctx->tcoef_dev, beta_dev, var_dev, pval_dev, P.D, half);
        } else
            hipLaunchKernelGGL(k_sweep_finish, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->spec_dev, ctx->syy_dev, ctx->tcoef_dev, beta_dev, var_dev, pval_dev, P.D);
        pg_prof_end(ctx);
        PG_HIP(ctx, hipGetLastError());
        return PG_OK;
    }
    // the matrix-core sweep unless an A/B run asks for one of the vector-ALU kernels
    const int cu = ctx->st_m + 1 + ctx->st_k;
    if (ms_fits(n, cu) && !std::getenv("POOLGEN_SWEEP_V1") && !std::getenv("POOLGEN_SWEEP_V2"))
        return launch_sweep_mfma<0>(ctx, P, ctx->st_cols, cu, PG_K_SWEEP);
    int64_t blocks = (P.D.ntiles + SW_WAVES - 1) / SW_WAVES;
    int mult = 8;
    if (const char *e = std::getenv("POOLGEN_SWEEP_GRID_MULT")) mult = std::max(1, std::atoi(e)); // experiments
    const int64_t cap = (int64_t)cus * mult;
    const int grid = (int)(blocks < cap ? blocks : cap);
    switch (ctx->st_cols) {
    case 2: return launch_sweep<2>(ctx, P, grid);
    case 3: return launch_sweep<3>(ctx, P, grid);
    case 4: return launch_sweep<4>(ctx, P, grid);
    case 6: return launch_sweep<6>(ctx, P, grid);
    case 8: return launch_sweep<8>(ctx, P, grid);
    case 12: return launch_sweep<12>(ctx, P, grid);
    case 16: return launch_sweep<16>(ctx, P, grid);
    case 24: return launch_sweep<24>(ctx, P, grid);
    case PG_MAX_SWEEP_COLS: return launch_sweep<PG_MAX_SWEEP_COLS>(ctx, P, grid);
    }
    return pg_fail(ctx, PG_ERR_STATE, "sweep: unexpected column count %d", ctx->st_cols);
}

extern "C" int pg_ols_kinship_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld,
                                  const double *Y, int k, double var_explained, int force_m,
                                  int *m_out, double *K_out, double *beta_dev, double *var_dev,
                                  double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    // ---- the lazy route: a caller that does not want K may not need it formed at all ----------------------------------------------
    // The rule of gwas/ols.rs:297-311 returns m = 0 as soon as lambda_1 / trace(K) >= x, and lambda_1 >= 1'K1 / n for any K: with
    // K = S / p that is (sum_l (sum_i g_li)^2 / n) / (sum_l sum_i g_li^2), two numbers the intercept-only sweep can form on the side.
    // One HBM-bound pass then gives the outputs of the m = 0 analysis (the sweep kernel's own: bit-identical to the two-pass
    // route); if the bound does not clear x by 1e-9 -- never on an uncentred kinship of real frequencies -- the full route below runs.
    if (!K_out && force_m < 0 && G_dev && Y && beta_dev && var_dev && pval_dev && p > 0 && n >= 3 && k >= 1 && k <= 15 && ld >= n &&
        (ld % 2) == 0 && (reinterpret_cast<uintptr_t>(G_dev) & 15) == 0 && ms_fits(n, 1 + k) && !std::getenv("POOLGEN_NO_LAZY_KINSHIP")) {
        rc = pg_covariates_set(ctx, n, nullptr, 0, Y, k);
        if (rc) return rc;
        const size_t lzbytes = sizeof(double) * 2 * (size_t)ctx->cus * 8 * 2; // a workgroup per CU (launch bounds), 8 waves each; twice that
        if (!ctx->lz_dev) PG_HIP(ctx, hipMalloc((void **)&ctx->lz_dev, lzbytes));
        PG_HIP(ctx, hipMemsetAsync(ctx->lz_dev, 0, lzbytes, ctx->stream));
        SweepArgs P;
        P.G = G_dev; P.W = ctx->W_dev; P.syy = ctx->syy_dev; P.tcoef = ctx->tcoef_dev;
        P.beta = beta_dev; P.var = var_dev; P.pval = pval_dev;
        std::memset(&P.Q, 0, sizeof P.Q);
        P.D.p = p; P.D.ld = ld; P.D.ntiles = 0;
        P.D.n = n; P.D.m1 = 1; P.D.k = k;
        P.D.tdf = ctx->tcoef_df; P.D.ntcoef = ctx->tcoef_len;
        P.D.dfe = (double)n - 2.0;
        P.D.tau = 1e-12;
        P.D.colmajor = 0;
        P.D.ss = nullptr;
        P.D.lz = ctx->lz_dev;
        rc = launch_sweep_mfma<2>(ctx, P, ctx->st_cols, 1 + k, PG_K_SWEEP);
        if (rc) return rc;
        rc = pg_pin_reserve(ctx, lzbytes);
        if (rc) return rc;
        PG_HIP(ctx, hipMemcpyAsync(ctx->pin, ctx->lz_dev, lzbytes, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double *hp = static_cast<const double *>(ctx->pin);
        double tot = 0.0, tr = 0.0;
        for (size_t w = 0; w < lzbytes / 16; ++w) { tot += hp[2 * w]; tr += hp[2 * w + 1]; }
        if (tr > 0.0 && std::isfinite(tot) && (tot / n) / tr >= var_explained + 1e-9) {
            if (m_out) *m_out = 0;
            ctx->lazy_taken = true;
            return PG_OK;
        }
        // (a NaN frequency, or a kinship whose leading share is not decided by the ones vector: the full route says what the reference says)
    }
    ctx->lazy_taken = false;
    if (ctx->S_n < n) {
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->S_dev) PG_HIP(ctx, hipFree(ctx->S_dev));
        ctx->S_dev = nullptr;
        ctx->S_n = 0;
        PG_HIP(ctx, hipMalloc((void **)&ctx->S_dev, sizeof(double) * n * n));
        ctx->S_n = n;
    }
    double *S_dev = ctx->S_dev;
    rc = pg_set_phenotypes(ctx, n, Y, k);
    if (rc) return rc;
    rc = pg_launch_kinship(ctx, G_dev, p, n, ld, S_dev, false, PG_K_KINSHIP, true);
    if (rc) return rc;
    rc = pg_kinship_set(ctx, S_dev, p, n, Y, k, var_explained, force_m, m_out, K_out, nullptr);
    if (rc) return rc;
    return pg_ols_sweep_dev(ctx, G_dev, p, n, ld, beta_dev, var_dev, pval_dev);
}

namespace {
__global__ void k_accumulate(double *__restrict__ acc, const double *__restrict__ x, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) acc[i] += x[i];
}
struct HostPathRes { // whatever pg_ols_kinship holds while it runs; released on every exit path
    double *Gd = nullptr, *out = nullptr, *S_acc = nullptr, *S_slab = nullptr;
    hipStream_t copy = nullptr;
    std::vector<hipEvent_t> ev;
    ~HostPathRes() {
        if (copy) { (void)hipStreamSynchronize(copy); (void)hipStreamDestroy(copy); }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        (void)hipFree(Gd); (void)hipFree(out); (void)hipFree(S_acc); (void)hipFree(S_slab);
    }
};
} // namespace

// Host-buffer form of ols_iter_with_kinship (what main.rs:285-291 hands over: the whole matrix in host memory).
// The link, not the GPU, sets its pace (16 GB at ~55 GB/s = 0.3 s against 10 ms of kernels), so the one thing worth
// doing is never to leave the link idle:
//   phase 1  G crosses the bus in SLABS on a copy stream (a pinned caller buffer is used as is; pageable pages are
//            pinned in place by the runtime, which reaches the link rate on this platform -- no second host copy is made
//            here) while the partial kinship of the slab that has already landed runs on the context's stream; the sums
//            are accumulated on the device in slab order; G stays resident in HBM (288 GB hold config 3 eighteen times);
//   n x n    eigen rule and basis on the accumulated sum (pg_kinship_set);
//   phase 2  the sweep runs slab by slab over the resident matrix and the results of slab s return over the bus while
//            slab s + 1 is swept.
// POOLGEN_HOST_SLAB_MB sets the slab size (default 256).
extern "C" int pg_ols_kinship(pg_ctx *ctx, const double *G, int64_t p, int n, int64_t ld,
                              const double *Y, int k, double var_explained, int force_m, int *m_out,
                              double *K_out, double *beta, double *var, double *pval) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G && Y && beta && var && pval && p > 0 && n >= 2 && k >= 1, "ols_kinship: bad arguments");
    PG_CHECK(ctx, ld >= n && (ld % 2) == 0, "ols_kinship: ld (%lld) must be even and >= n (%d)", (long long)ld, n);
    PG_HIP(ctx, hipSetDevice(ctx->device));
    long slab_mb = 256;
    if (const char *e = std::getenv("POOLGEN_HOST_SLAB_MB")) slab_mb = std::max(1L, std::atol(e));
    int64_t slab_loci = ((int64_t)slab_mb << 20) / (ld * 8);
    slab_loci = std::max<int64_t>(1024, slab_loci / 1024 * 1024); // a multiple of the sweep's super-row tiles
    const int nslab = (int)((p + slab_loci - 1) / slab_loci);
    HostPathRes R;
    const size_t gbytes = (size_t)p * ld * sizeof(double);
    const size_t ocnt = (size_t)p * k;
    if (hipMalloc((void **)&R.Gd, gbytes) != hipSuccess || hipMalloc((void **)&R.out, 3 * ocnt * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&R.S_acc, sizeof(double) * n * n) != hipSuccess || hipMalloc((void **)&R.S_slab, sizeof(double) * n * n) != hipSuccess)
        return pg_fail(ctx, PG_ERR_HIP, "ols_kinship: out of device memory (%.1f GB for the matrix)", gbytes / 1e9);
    PG_HIP(ctx, hipStreamCreateWithFlags(&R.copy, hipStreamNonBlocking));
    R.ev.resize((size_t)2 * nslab + 1, nullptr);
    for (auto &e : R.ev) PG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    PG_HIP(ctx, hipMemsetAsync(R.S_acc, 0, sizeof(double) * n * n, ctx->stream));
    int rc = pg_set_phenotypes(ctx, 0, nullptr, 0); // the fused intercept-only sums belong to the resident path
    if (rc) return rc;
    // ---- phase 1: H2D of slab s + 1 || partial kinship of slab s ------------------------------------------------
    auto h2d = [&](int sidx) -> hipError_t {
        const int64_t lo = (int64_t)sidx * slab_loci, cnt = std::min(slab_loci, p - lo);
        hipError_t e = hipMemcpyAsync(R.Gd + lo * ld, G + lo * ld, (size_t)cnt * ld * sizeof(double), hipMemcpyHostToDevice, R.copy);
        if (e != hipSuccess) return e;
        return hipEventRecord(R.ev[sidx], R.copy);
    };
    PG_HIP(ctx, h2d(0));
    for (int sidx = 0; sidx < nslab; ++sidx) {
        const int64_t lo = (int64_t)sidx * slab_loci, cnt = std::min(slab_loci, p - lo);
        PG_HIP(ctx, hipStreamWaitEvent(ctx->stream, R.ev[sidx], 0));
        rc = pg_launch_kinship(ctx, R.Gd + lo * ld, cnt, n, ld, R.S_slab, false, PG_K_KINSHIP, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_accumulate, dim3((n * n + 255) / 256), dim3(256), 0, ctx->stream, R.S_acc, R.S_slab, n * n);
        PG_HIP(ctx, hipGetLastError());
        if (sidx + 1 < nslab) PG_HIP(ctx, h2d(sidx + 1)); // queued behind nothing but the previous slab's copy
    }
    // ---- the n x n step ---------------------------------------------------------------------------------------------
    rc = pg_kinship_set(ctx, R.S_acc, p, n, Y, k, var_explained, force_m, m_out, K_out, nullptr);
    if (rc) return rc;
    // ---- phase 2: sweep of slab s + 1 || D2H of the results of slab s -------------------------------------------
    double *ob = R.out, *ov = R.out + ocnt, *op = R.out + 2 * ocnt;
    for (int sidx = 0; sidx < nslab; ++sidx) {
        const int64_t lo = (int64_t)sidx * slab_loci, cnt = std::min(slab_loci, p - lo);
        rc = pg_ols_sweep_dev(ctx, R.Gd + lo * ld, cnt, n, ld, ob + lo * k, ov + lo * k, op + lo * k);
        if (rc) return rc;
        hipEvent_t done = R.ev[(size_t)nslab + sidx];
        PG_HIP(ctx, hipEventRecord(done, ctx->stream));
        PG_HIP(ctx, hipStreamWaitEvent(R.copy, done, 0));
        const size_t bytes = (size_t)cnt * k * sizeof(double);
        PG_HIP(ctx, hipMemcpyAsync(beta + lo * k, ob + lo * k, bytes, hipMemcpyDeviceToHost, R.copy));
        PG_HIP(ctx, hipMemcpyAsync(var + lo * k, ov + lo * k, bytes, hipMemcpyDeviceToHost, R.copy));
        PG_HIP(ctx, hipMemcpyAsync(pval + lo * k, op + lo * k, bytes, hipMemcpyDeviceToHost, R.copy));
    }
    PG_HIP(ctx, hipStreamSynchronize(R.copy));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// gp::ols, n < p branch (gp/ols.rs:47-72): b = X^T pinv(X X^T) y over the training rows.
// ---------------------------------------------------------------------------------------------
template <int C>
static int launch_gp_beta(pg_ctx *ctx, const double *G, const double *W, double *out, const SweepDims &D, int grid) {
    const size_t shmem = (size_t)SW_WAVES * SW_TILE * sizeof(double);
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_gp_beta<C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    pg_prof_begin(ctx, PG_K_GP_BETA);
    hipLaunchKernelGGL(k_gp_beta<C>, dim3(grid), dim3(SW_THREADS), shmem, ctx->stream, G, W, out, D);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

template <int C>
static int launch_gp_beta_lds(pg_ctx *ctx, const double *G, const double *W, double *out, const SweepDims &D, int wdoubles) {
    const size_t shmem = ((size_t)wdoubles + (size_t)SW_WAVES * SW_TILE) * sizeof(double);
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_gp_beta_lds<C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int64_t blocks = (D.ntiles + SW_WAVES - 1) / SW_WAVES;
    const int grid = (int)std::min<int64_t>(blocks, (int64_t)ctx->cus); // one block per CU: Z occupies most of its LDS
    pg_prof_begin(ctx, PG_K_GP_BETA);
    hipLaunchKernelGGL(k_gp_beta_lds<C>, dim3(grid), dim3(SW_THREADS), shmem, ctx->stream, G, W, out, D, wdoubles);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

// ---- internal pieces of gp::ols shared with the ridge path (pg_gp.hip) ---------------------------------
// V (r x k) = pinv(A) Y_rows with A the principal sub-block `rows` of the full-data X X^T (n x n, host)
int pg_gp_subset_solve(const double *xxt, int n, const double *Y, int k, const int64_t *rows, int r, double *V) {
    std::vector<double> A((size_t)r * r), Ysub((size_t)r * k);
    for (int a = 0; a < r; ++a) {
        for (int b = 0; b < r; ++b) A[(size_t)a * r + b] = xxt[(size_t)rows[a] * n + rows[b]];
        for (int j = 0; j < k; ++j) Ysub[(size_t)a * k + j] = Y[(size_t)rows[a] * k + j];
    }
    return pg_pinv_solve_sym(A.data(), r, Ysub.data(), k, V);
}

// out (p x ncol, device) = G Z for a host Z (n x ncol row-major): the slopes of `ncol` fits in ONE pass over G
int pg_gp_beta_cols(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Z_host, int ncol,
                    double *out_dev, int colmajor, double *ss_out_dev) {
    const int cols = round_cols(ncol);
    if (cols < 0) return pg_fail(ctx, PG_ERR_UNSUPPORTED, "gp: at most %d coefficient columns per pass", PG_MAX_SWEEP_COLS);
    const int n_even = (n + 1) & ~1;
    std::vector<double> Z((size_t)n_even * cols, 0.0);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < ncol; ++c) Z[(size_t)i * cols + c] = Z_host[(size_t)i * ncol + c];
    const size_t zbytes = Z.size() * sizeof(double);
    if (zbytes > ctx->W_cap) {
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->W_dev) PG_HIP(ctx, hipFree(ctx->W_dev));
        ctx->W_dev = nullptr; ctx->W_cap = 0;
        PG_HIP(ctx, hipMalloc((void **)&ctx->W_dev, zbytes));
        ctx->W_cap = zbytes;
    }
    ctx->st_m = -1; // the regression state in W_dev is gone
    ctx->st_Y.clear();
    PG_HIP(ctx, hipMemcpyAsync(ctx->W_dev, Z.data(), zbytes, hipMemcpyHostToDevice, ctx->stream));
    SweepDims D;
    std::memset(&D, 0, sizeof D);
    D.p = p; D.ld = ld; D.ntiles = (p + 63) / 64; D.n = n; D.k = ncol; D.colmajor = colmajor;
    D.ss = ss_out_dev; // (only the scalar-operand kernel below writes it)
    D.lz = nullptr;
    int64_t blocks = (D.ntiles + SW_WAVES - 1) / SW_WAVES;
    const int64_t cap = (int64_t)ctx->cus * 8;
    const int grid = (int)(blocks < cap ? blocks : cap);
    int rc;
    // the matrix-core kernel of the sweep in its products-only mode: every shape, one read of G at the sweep's rate
    if (ms_fits(n, ncol) && (ld % 2) == 0 && (reinterpret_cast<uintptr_t>(G_dev) & 15) == 0 && !std::getenv("POOLGEN_GP_BETA_OLD")) {
        SweepArgs P;
        P.G = G_dev; P.W = ctx->W_dev; P.syy = nullptr; P.tcoef = nullptr;
        P.beta = out_dev; P.var = nullptr; P.pval = nullptr;
        P.D = D;
        rc = launch_sweep_mfma<1>(ctx, P, cols, ncol, PG_K_GP_BETA);
        if (rc) return rc;
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // Z is stack-owned
        return PG_OK;
    }
    // (the older forms, kept for A/B timing: POOLGEN_GP_BETA_OLD=1)
    // the folds' slopes of a CV repetition (column-major, up to 16 columns): the MFMA form over an LDS-staged tile
    const int zrows = (n + SW_CH - 1) / SW_CH * SW_CH;
    const size_t mfma_lds = ((size_t)zrows * 16 + (size_t)SW_WAVES * MB_TILE) * sizeof(double);
    if (!ss_out_dev && colmajor && ncol >= 5 && ncol <= 16 && mfma_lds <= 150 * 1024 && !std::getenv("POOLGEN_GP_BETA_VALU")) {
        std::vector<double> Z16((size_t)zrows * 16, 0.0);
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < ncol; ++c) Z16[(size_t)i * 16 + c] = Z_host[(size_t)i * ncol + c];
        const size_t zb = Z16.size() * sizeof(double);
        if (zb > ctx->W_cap) {
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->W_dev) PG_HIP(ctx, hipFree(ctx->W_dev));
            ctx->W_dev = nullptr; ctx->W_cap = 0;
            PG_HIP(ctx, hipMalloc((void **)&ctx->W_dev, zb));
            ctx->W_cap = zb;
        }
        PG_HIP(ctx, hipMemcpyAsync(ctx->W_dev, Z16.data(), zb, hipMemcpyHostToDevice, ctx->stream));
        PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_gp_beta_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)mfma_lds));
        const int64_t nblocks = (D.ntiles + SW_WAVES - 1) / SW_WAVES;
        const int g2 = (int)std::min<int64_t>(nblocks, (int64_t)ctx->cus);
        pg_prof_begin(ctx, PG_K_GP_BETA);
        hipLaunchKernelGGL(k_gp_beta_mfma, dim3(g2), dim3(SW_THREADS), mfma_lds, ctx->stream, G_dev, ctx->W_dev, out_dev, D, zrows);
        pg_prof_end(ctx);
        PG_HIP(ctx, hipGetLastError());
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // Z16 is stack-owned
        return PG_OK;
    }
    // many columns x many pools: Z no longer fits the scalar cache -> the LDS-resident variant (see k_gp_beta_lds)
    const int wdoubles = n_even * cols;
    const size_t lds_need = ((size_t)wdoubles + (size_t)SW_WAVES * SW_TILE) * sizeof(double);
    if (!ss_out_dev && cols >= 6 && (cols % 2) == 0 && cols <= 24 && (size_t)wdoubles * sizeof(double) > 12288 && lds_need <= 150 * 1024 &&
        !std::getenv("POOLGEN_GP_BETA_SCALAR")) {
        switch (cols) {
        case 6: rc = launch_gp_beta_lds<6>(ctx, G_dev, ctx->W_dev, out_dev, D, wdoubles); break;
        case 8: rc = launch_gp_beta_lds<8>(ctx, G_dev, ctx->W_dev, out_dev, D, wdoubles); break;
        case 12: rc = launch_gp_beta_lds<12>(ctx, G_dev, ctx->W_dev, out_dev, D, wdoubles); break;
        case 16: rc = launch_gp_beta_lds<16>(ctx, G_dev, ctx->W_dev, out_dev, D, wdoubles); break;
        default: rc = launch_gp_beta_lds<24>(ctx, G_dev, ctx->W_dev, out_dev, D, wdoubles); break;
        }
        if (rc) return rc;
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // Z is stack-owned
        return PG_OK;
    }
    switch (cols) {
    case 2: rc = launch_gp_beta<2>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 3: rc = launch_gp_beta<3>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 4: rc = launch_gp_beta<4>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 6: rc = launch_gp_beta<6>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 8: rc = launch_gp_beta<8>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 12: rc = launch_gp_beta<12>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 16: rc = launch_gp_beta<16>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    case 24: rc = launch_gp_beta<24>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    default: rc = launch_gp_beta<PG_MAX_SWEEP_COLS>(ctx, G_dev, ctx->W_dev, out_dev, D, grid); break;
    }
    if (rc) return rc;
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // Z is stack-owned
    return PG_OK;
}

extern "C" int pg_gp_ols_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y,
                             int k, const int64_t *row_idx, int n_rows, const double *XXt_host_or_null,
                             double *beta_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G_dev && Y && row_idx && beta_dev && p > 0 && n >= 1 && k >= 1 && n_rows >= 1 && n_rows <= n,
             "gp_ols: bad arguments");
    PG_CHECK(ctx, ld >= n && (ld % 2) == 0, "gp_ols: ld must be even and >= n");
    if (k > 8) return pg_fail(ctx, PG_ERR_UNSUPPORTED, "gp_ols: at most 8 traits per call");
    for (int a = 0; a < n_rows; ++a) PG_CHECK(ctx, row_idx[a] >= 0 && row_idx[a] < n, "gp_ols: row index out of range");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    if ((int64_t)n >= p + 1) {
        // The tall branch (gp/ols.rs:72-99, taken when x.nrows() >= x.ncols(): at most n - 1 loci, i.e. the reference's own 5 x 3
        // test, never a pool-seq matrix): b = pinv(X'X over the training rows) X' y.  (1 + p)^2 <= n^2 numbers: the host's,
        // not a GPU problem.  pinv as in the wide branch (helpers.rs:463-482).
        const int P = (int)p + 1;
        std::vector<double> Gh((size_t)p * ld);
        PG_HIP(ctx, hipMemcpyAsync(Gh.data(), G_dev, sizeof(double) * (size_t)p * ld, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        auto X = [&](int64_t i, int c) { return c == 0 ? 1.0 : Gh[(size_t)(c - 1) * ld + i]; };
        std::vector<double> xtx((size_t)P * P), pinv((size_t)P * P), T((size_t)P * n_rows), b((size_t)P * k);
        for (int a = 0; a < P; ++a)
            for (int c = 0; c < P; ++c) {
                double x = 0.0;
                for (int i = 0; i < n_rows; ++i) x += X(row_idx[i], a) * X(row_idx[i], c);
                xtx[(size_t)a * P + c] = x;
            }
        if (pg_pinv_sym(xtx.data(), P, pinv.data()) != 0) return pg_fail(ctx, PG_ERR_INVALID, "gp_ols: pinv failed");
        for (int a = 0; a < P; ++a) // (pinv X') y, in the reference's order of products
            for (int i = 0; i < n_rows; ++i) {
                double x = 0.0;
                for (int c = 0; c < P; ++c) x += pinv[(size_t)a * P + c] * X(row_idx[i], c);
                T[(size_t)a * n_rows + i] = x;
            }
        for (int a = 0; a < P; ++a)
            for (int j = 0; j < k; ++j) {
                double x = 0.0;
                for (int i = 0; i < n_rows; ++i) x += T[(size_t)a * n_rows + i] * Y[(size_t)row_idx[i] * k + j];
                b[(size_t)a * k + j] = x;
            }
        PG_HIP(ctx, hipMemcpyAsync(beta_dev, b.data(), sizeof(double) * (size_t)P * k, hipMemcpyHostToDevice, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // b is stack-owned
        return PG_OK;
    }
    std::vector<double> full((size_t)n * n);
    if (XXt_host_or_null) {
        std::memcpy(full.data(), XXt_host_or_null, sizeof(double) * n * n);
    } else {
        if (ctx->S_n < n) {
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->S_dev) PG_HIP(ctx, hipFree(ctx->S_dev));
            ctx->S_dev = nullptr; ctx->S_n = 0;
            PG_HIP(ctx, hipMalloc((void **)&ctx->S_dev, sizeof(double) * n * n));
            ctx->S_n = n;
        }
        int rc = pg_launch_kinship(ctx, G_dev, p, n, ld, ctx->S_dev, true, PG_K_GP_XXT);
        if (rc) return rc;
        PG_HIP(ctx, hipMemcpyAsync(full.data(), ctx->S_dev, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // every training subset's X X^T is a principal sub-block of the full-data one
    const int r = n_rows;
    std::vector<double> V((size_t)r * k);
    if (pg_gp_subset_solve(full.data(), n, Y, k, row_idx, r, V.data()) != 0) return pg_fail(ctx, PG_ERR_INVALID, "gp_ols: pinv failed");
    std::vector<double> Z((size_t)n * k, 0.0), b0(k, 0.0);
    for (int a = 0; a < r; ++a)
        for (int j = 0; j < k; ++j) {
            Z[(size_t)row_idx[a] * k + j] = V[(size_t)a * k + j];
            b0[j] += V[(size_t)a * k + j]; // intercept column of X is all ones
        }
    PG_HIP(ctx, hipMemcpyAsync(beta_dev, b0.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    int rc = pg_gp_beta_cols(ctx, G_dev, p, n, ld, Z.data(), k, beta_dev + k); // rows 1..p
    if (rc) return rc;
    return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// gp::ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199): the proxy coefficients of the
// *_with_iterative_proxy_norms models.  Per locus the last coefficient of y ~ [1 | PC1 | g] on the training
// pools, PC1 = leading eigenvector of the "kinship" of the training pools -- the very sweep of
// ols_iter_with_kinship with one covariate, run on the training pools' columns of G.
// ---------------------------------------------------------------------------------------------
namespace {

__global__ void k_gather_pools(const double *__restrict__ G, int64_t p, int64_t ld, const int32_t *__restrict__ rows,
                               int nr, int64_t ld2, double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p * ld2) return;
    const int64_t l = idx / ld2;
    const int i = (int)(idx - l * ld2);
    out[idx] = i < nr ? G[l * ld + rows[i]] : 0.0;
}

struct ProxyFix { double a0[8]; };
// A locus whose frequencies are the same in every training pool is collinear with the intercept; the reference's
// least_squares (LAPACK gelsd, :193) returns the minimum-norm solution there: with y ~ a0 + b0 PC1 the fit of the
// other two columns, the set of solutions is a + c d = a0, and the shortest one has d = c a0 / (1 + c^2).
__global__ void k_proxy_fix_constant(const double *__restrict__ Gs, int64_t p, int nr, int64_t ld2, ProxyFix F, int k,
                                     double *__restrict__ beta) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= p) return;
    if (!isnan(beta[l * k])) return;
    const double c = Gs[l * ld2];
    for (int i = 1; i < nr; ++i)
        if (Gs[l * ld2 + i] != c) return; // singular for another reason: stays NaN
    for (int j = 0; j < k; ++j) beta[l * k + j] = c * F.a0[j] / (1.0 + c * c);
}

} // namespace

extern "C" int pg_gp_proxy_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y, int k,
                               const int64_t *row_idx, int n_rows, const double *XXt_host_or_null, double *proxy_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G_dev && Y && row_idx && proxy_dev && p >= 2 && n >= 4 && k >= 1 && k <= 8 && n_rows >= 4 && n_rows <= n,
             "gp_proxy: bad arguments");
    PG_CHECK(ctx, ld >= n, "gp_proxy: ld must be >= n");
    for (int a = 0; a < n_rows; ++a) PG_CHECK(ctx, row_idx[a] >= 0 && row_idx[a] < n, "gp_proxy: row index out of range");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int nr = n_rows;
    std::vector<double> T((size_t)n * n), glast(n);
    if (XXt_host_or_null) std::memcpy(T.data(), XXt_host_or_null, sizeof(double) * n * n);
    else {
        double *S = nullptr;
        PG_HIP(ctx, hipMalloc((void **)&S, sizeof(double) * n * n));
        int rc = pg_gp_xxt_dev(ctx, G_dev, p, n, ld, S);
        if (rc == PG_OK && (hipMemcpyAsync(T.data(), S, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                            hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = pg_fail(ctx, PG_ERR_HIP, "gp_proxy: D2H failed");
        (void)hipFree(S);
        if (rc) return rc;
    }
    PG_HIP(ctx, hipMemcpyAsync(glast.data(), G_dev + (p - 1) * ld, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // The reference centres columns 0..P-2 of x = [1 | G^T] (the intercept IS among them, the last locus is NOT, :115),
    // each by its mean over the FIRST n_rows rows of x (not over row_idx, :124-129), then takes X_c X_c^T over the
    // training rows (:132-140).  With T = x x^T over those columns (the full X X^T minus the last locus' outer
    // product): K[a][b] = T[ra][rb] - u[ra] - u[rb] + c,  u[r] = mean_{i<nr} T[i][r],  c = mean_{i,i'<nr} T[i][i'].
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) T[(size_t)i * n + j] -= glast[i] * glast[j];
    std::vector<double> u(n, 0.0), K((size_t)nr * nr);
    double c = 0.0;
    for (int r = 0; r < n; ++r) {
        double s = 0.0;
        for (int i = 0; i < nr; ++i) s += T[(size_t)i * n + r];
        u[r] = s / (double)nr;
    }
    for (int i = 0; i < nr; ++i) c += u[i];
    c /= (double)nr;
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nr; ++b)
            K[(size_t)a * nr + b] = T[(size_t)row_idx[a] * n + row_idx[b]] - u[row_idx[a]] - u[row_idx[b]] + c;
    // eigen_vectors column 0 (:141, :177): the leading one, under the same reading of the LAPACK order as
    // ols_with_covariate (gwas/ols.rs:296)
    std::vector<double> evals(nr), ev(nr);
    if (pg_sym_eig_top(K.data(), nr, 1, evals.data(), ev.data()) != 0)
        return pg_fail(ctx, PG_ERR_INVALID, "gp_proxy: eigen-decomposition failed");
    std::vector<double> Ys((size_t)nr * k), ymean(k, 0.0);
    for (int a = 0; a < nr; ++a)
        for (int j = 0; j < k; ++j) { Ys[(size_t)a * k + j] = Y[(size_t)row_idx[a] * k + j]; ymean[j] += Ys[(size_t)a * k + j]; }
    for (int j = 0; j < k; ++j) ymean[j] /= (double)nr; // row 0 (:170-172)
    // the training pools' columns, compacted (the sweep reads whole rows of pools) -- unless they are all pools in order
    bool identity = nr == n && (ld % 2) == 0 && (reinterpret_cast<uintptr_t>(G_dev) & 15) == 0;
    for (int a = 0; a < nr && identity; ++a) identity = row_idx[a] == a;
    const int64_t ld2 = identity ? ld : nr + (nr & 1);
    double *Gs = nullptr, *scratch = nullptr;
    int32_t *rows_dev = nullptr;
    std::vector<int32_t> rows32(nr);
    for (int a = 0; a < nr; ++a) rows32[a] = (int32_t)row_idx[a];
    auto cleanup = [&] { if (!identity) (void)hipFree(Gs); (void)hipFree(scratch); (void)hipFree(rows_dev); };
    if ((!identity && hipMalloc((void **)&Gs, sizeof(double) * (size_t)p * ld2) != hipSuccess) ||
        hipMalloc((void **)&scratch, sizeof(double) * (size_t)p * k) != hipSuccess ||
        hipMalloc((void **)&rows_dev, sizeof(int32_t) * nr) != hipSuccess) {
        cleanup();
        return pg_fail(ctx, PG_ERR_HIP, "gp_proxy: out of device memory");
    }
    if (hipMemcpyAsync(rows_dev, rows32.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        cleanup();
        return pg_fail(ctx, PG_ERR_HIP, "gp_proxy: H2D failed");
    }
    if (identity) Gs = const_cast<double *>(G_dev);
    else
        hipLaunchKernelGGL(k_gather_pools, dim3((unsigned)(((size_t)p * ld2 + 255) / 256)), dim3(256), 0, ctx->stream, G_dev, p, ld,
                           rows_dev, nr, ld2, Gs);
    int rc = pg_covariates_set(ctx, nr, ev.data(), 1, Ys.data(), k);
    if (rc == PG_OK) rc = pg_ols_sweep_dev(ctx, Gs, p, nr, ld2, proxy_dev + k, scratch, scratch);
    if (rc == PG_OK) {
        // y ~ a0 + b0 PC1 for the constant loci
        ProxyFix F{};
        double s1 = 0.0, s2 = 0.0;
        for (int a = 0; a < nr; ++a) { s1 += ev[a]; s2 += ev[a] * ev[a]; }
        const double det = (double)nr * s2 - s1 * s1;
        for (int j = 0; j < k; ++j) {
            double sy = 0.0, sey = 0.0;
            for (int a = 0; a < nr; ++a) { sy += Ys[(size_t)a * k + j]; sey += ev[a] * Ys[(size_t)a * k + j]; }
            F.a0[j] = (s2 * sy - s1 * sey) / det;
        }
        hipLaunchKernelGGL(k_proxy_fix_constant, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, Gs, p, nr, ld2, F, k,
                           proxy_dev + k);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(proxy_dev, ymean.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = pg_fail(ctx, PG_ERR_HIP, "gp_proxy: closing step failed");
    }
    cleanup();
    return rc;
}
