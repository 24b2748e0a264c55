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
// Every locus' counts are read ONCE by k_locus_stream (a lane streams whole 128-byte lines of its own loci through a
// wave-private LDS ring, see there), which decides the filter, takes the operator's sums speculatively with every
// candidate allele in play and closes biallelic loci in place; only loci whose dropped alleles carry reads, or (ols_iter,
// pearson_corr) that keep three or more alleles, are listed for k_locus_second, which redoes them from the counts over
// the SURVIVING alleles (row sums change) and closes them.  The helpers below up to `Sums` serve that second pass.
#include "pg_common.h"
#include "pg_stats_device.h"
#include <cmath>
#include <type_traits>
#include <vector>

namespace {

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

#ifndef LO_WAVES_DEF
#define LO_WAVES_DEF 4   // waves per workgroup of the streaming kernels
#endif
#ifndef LO_BLOCKS_DEF
#define LO_BLOCKS_DEF 2  // workgroups per CU they are compiled for
#endif
constexpr int LO_WAVES = LO_WAVES_DEF;
constexpr int LO_THREADS = 64 * LO_WAVES;
constexpr int LO_CHP = 8;                 // pools per stage (second pass)
constexpr int LO_ROWB = LO_CHP * 24;      // 192 bytes of counts per locus per stage
constexpr int LO_PITCH = LO_ROWB + 16;    // 208: odd number of 16-byte slots
constexpr int LO_TILEB = 64 * LO_PITCH;   // bytes per wave
constexpr int NA = 6;                     // sync alleles
constexpr int MAXK = 2;                   // traits per launch (the host loops over trait pairs)

enum { OP_OLS = 0, OP_PEARSON = 1, OP_CHISQ = 2, OP_LOAD = 3 }; // OP_LOAD: filter decisions + column order only (the loader)

struct LocusParams {
    int64_t L;
    int n, k;        // k = traits handled by this launch (<= MAXK)
    int k_total, t0; // output layout: trait t0 + tt of k_total
    int remove_ns;
    int pshift;      // unit_slot's interleave (0 since round 3: records and flags are indexed by the locus itself)
    int sort_desc;   // OP_LOAD: order the surviving alleles by decreasing column sum (--keep-p-minus-1)
    double min_cov, maf, max_miss;
    int tdf, ntcoef;     // t-test degrees of freedom (OLS: n-1, Pearson: n-2)
    double syy[MAXK];    // OLS: sum of centred y^2
    double sy[MAXK];     // OLS: sum of centred y (~0)
    // PEARSON: sum y, sum y^2 (fma chain) and the number of pools, y shifted by its first value, in pool order -- what the running
    // sums over the complete pairs come to when every pair is complete; y_complete = 0 (a NaN phenotype) lists every locus
    double py[MAXK], pyy[MAXK], pn[MAXK];
    int y_complete;
    // the streaming pass decides q < maf on a cheaper evaluation of q (see k_locus_stream); |q - threshold| <= qband sends the
    // locus through the exact evaluation
    double qband;
};
constexpr int LO_NB = 5;   // groups of the second pass' list: one per number of surviving alleles, 2 .. 6
// words of the second pass' counter block: [0, LO_NB) loci per group | complaint flag | length of the list | (pad) | [SC_CURSOR, +LO_NB) the
// sort's cursors
constexpr int SC_COMPLAINT = LO_NB, SC_LIST = LO_NB + 1, SC_DIRTY = LO_NB + 2, SC_CURSOR = LO_NB + 3, SC_WORDS = 2 * LO_NB + 3;
// (SC_DIRTY: k_ols_rows counts the loci that carry reads of alleles the filter drops, or keep three alleles or more)
constexpr int LO_GROUP_FROM = 1 << 17; // lists from this length on are grouped by the number of survivors (two more launches)

typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

// Output layout (include/poolgen_hip.h): SLOT-MAJOR -- element (slot r, locus l) of allele_ids / mean_freq at r * L + l, of
// stat / pval at (r * L + l) * k + trait.  A locus that emits one row (the biallelic case) touches slot 0 only, so the streaming
// pass writes 32 bytes per locus instead of the 144 of a locus-major [L][PG_MAX_OUT] layout (measured: -4 % of the pass).
#define PG_OIX(l, r) ((size_t)(r) * (size_t)P.L + (size_t)(l))

// ---- staging: HBM -> registers -> wave-private LDS tile ------------------------------------------
// A locus row of a stage is 192 bytes = 12 pieces of 16 B (24 of 8 B when rows are only 8-byte
// aligned, and always for the last, partial stage): 48 lanes cover 4 (2) loci per instruction, so that
// every address is "lane base + r * constant".  The loads of a stage land in 64 registers per lane and
// are written to the tile one stage later: they are in flight while the previous stage is computed.
// The partial stage moves 8-byte pieces whose offset is clamped INSIDE the row (no read past the end
// of the batch) and which land at that same offset in the tile.
struct StageRegs { uint32_t v[64]; };

template <int PBE> struct StageGeom {
    static constexpr int PPR = LO_ROWB / PBE;  // pieces per locus row: 12 or 24
    static constexpr int LPI = 48 / PPR;       // loci per wave instruction: 4 or 2
    static constexpr int NI = 64 / LPI;        // instructions per stage: 16 or 32
};

// rowsel(r_locus) -> global locus index of tile row r_locus (0..63)
template <int PBE, bool CLAMP, typename RowSel>
__device__ __forceinline__ void stage_load(StageRegs &S, const uint32_t *__restrict__ counts, int n, int pool0,
                                           int np, int lane, RowSel rowsel) {
    using G = StageGeom<PBE>;
    if (lane < 48) {
        const int64_t rowb = (int64_t)n * 24;
        const int sub = lane / G::PPR;
        const int pc = lane - sub * G::PPR;
        int off = pc * PBE;
        if (CLAMP) {
            const int valid = np * 24;
            off = off < valid ? off : valid - PBE;
        }
        const char *gbase = reinterpret_cast<const char *>(counts) + (int64_t)pool0 * 24 + off;
#pragma unroll
        for (int r = 0; r < G::NI; ++r) {
            const int64_t l = rowsel(G::LPI * r + sub);
            if (PBE == 16) {
                const uint4_t x = *reinterpret_cast<const uint4_t *>(gbase + l * rowb);
                S.v[4 * r] = x.x; S.v[4 * r + 1] = x.y; S.v[4 * r + 2] = x.z; S.v[4 * r + 3] = x.w;
            } else {
                const uint2_t x = *reinterpret_cast<const uint2_t *>(gbase + l * rowb);
                S.v[2 * r] = x.x; S.v[2 * r + 1] = x.y;
            }
        }
    }
}

template <int PBE, bool CLAMP>
__device__ __forceinline__ void stage_store(const StageRegs &S, char *tile, int np, int lane) {
    using G = StageGeom<PBE>;
    if (lane < 48) {
        const int sub = lane / G::PPR;
        const int pc = lane - sub * G::PPR;
        int off = pc * PBE;
        if (CLAMP) {
            const int valid = np * 24;
            off = off < valid ? off : valid - PBE;
        }
        char *tbase = tile + sub * LO_PITCH + off;
#pragma unroll
        for (int r = 0; r < G::NI; ++r) {
            if (PBE == 16) {
                uint4_t x;
                x.x = S.v[4 * r]; x.y = S.v[4 * r + 1]; x.z = S.v[4 * r + 2]; x.w = S.v[4 * r + 3];
                *reinterpret_cast<uint4_t *>(tbase + r * (G::LPI * LO_PITCH)) = x;
            } else {
                uint2_t x;
                x.x = S.v[2 * r]; x.y = S.v[2 * r + 1];
                *reinterpret_cast<uint2_t *>(tbase + r * (G::LPI * LO_PITCH)) = x;
            }
        }
    }
}

// IEEE-correct c / rs for a whole pool from ONE reciprocal.  This is the arithmetic hipcc itself
// emits for an fp64 division of normal-range operands (v_rcp_f64, two Newton steps on the
// reciprocal, q0 = c * r, one fused residual, one fused correction -- the v_div_scale / v_div_fixup
// wrappers only act on over-/underflowing operands, which counts and coverages are not), with the
// reciprocal shared by the (up to six) alleles of the pool instead of being recomputed per allele.
// The quotients are therefore bit-identical to `c / rs`; tests/test_gpu_locus_ops.py pins that
// through the bit-exact mean frequencies and filter decisions.  rs == 0 gives NaN (0 * inf), which
// is what to_frequencies produces for an uncovered pool (sync.rs:176-183).
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
constexpr int REC_DOUBLES = 42; // the largest compact record: pearson, 5 outputs x 2 traits (see emit_record)
// Records (of the second pass) and flags live in groups of 64 slots: field f of slot s at
// rec[(s / 64) * REC_DOUBLES * 64 + f * 64 + s % 64], so that a wave's stores are whole 512-byte runs.  slot = unit_slot(l, pshift);
// with pshift = 0 (every caller since round 3) the slot is the locus.
__device__ __forceinline__ int64_t unit_slot(int64_t l, int pshift) {
    const int64_t g = l >> (6 + pshift);
    const int within = (int)(l - (g << (6 + pshift)));
    const int c = within & ((1 << pshift) - 1);
    return (((g << pshift) + c) << 6) + (within >> pshift);
}
__device__ __forceinline__ size_t rec_base(int64_t slot) {
    return (size_t)(slot >> 6) * (size_t)(REC_DOUBLES * 64) + (size_t)(slot & 63);
}
constexpr int FLAG_ALIVE = 1, FLAG_SECOND = 1 << 7; // bits 1..6: surviving alleles

// The operator's running sums of one locus over the alleles "in play" (NJ of them; allele id of slot
// jj is aj(jj)).  Everything is accumulated sequentially in pool order.
template <int OP, int NJ, int K>
struct Sums {
    double cs[NJ];                 // NaN-ignoring column sums (sort key, mean frequency); plain adds
    double xx[NJ * (NJ + 1) / 2];  // OLS: sum f_a f_b (a <= b); CHISQ: first NJ entries = sum f_j^2 / rowsum
    double xy[NJ * K];          // OLS: sum f_j y_t; PEARSON: sum x y over complete pairs
    double px[NJ * K], pxx[NJ * K], py[K], pyy[K], pn[K], shx[NJ];
    double total;
    bool shset;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { cs[j] = 0.0; shx[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < NJ * (NJ + 1) / 2; ++j) xx[j] = 0.0;
#pragma unroll
        for (int j = 0; j < NJ * K; ++j) { xy[j] = 0.0; px[j] = 0.0; pxx[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < K; ++j) { py[j] = 0.0; pyy[j] = 0.0; pn[j] = 0.0; }
        total = 0.0;
        shset = false;
    }
    static __device__ __forceinline__ constexpr int trin(int a, int b) { return a * NJ - a * (a - 1) / 2 + (b - a); }
    // f: frequencies of this pool, all 0 when the pool is uncovered (rowok false).  The reference has NaN
    // there: its NaN-ignoring sums (cs, the Pearson sums) skip the pool, which adding 0 / `ok` does
    // too; its plain sums (OLS, chi-square) become NaN, which store_record does from `poisoned`.
    __device__ __forceinline__ void add_pool(const double (&f)[NJ], bool rowok, const double *__restrict__ Yrow) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) cs[j] = cs[j] + f[j];
        if (OP == OP_OLS) {
#pragma unroll
            for (int a = 0; a < NJ; ++a)
#pragma unroll
                for (int b = a; b < NJ; ++b) xx[trin(a, b)] = fma(f[a], f[b], xx[trin(a, b)]);
#pragma unroll
            for (int tt = 0; tt < K; ++tt) {
                const double y = Yrow[tt];
#pragma unroll
                for (int j = 0; j < NJ; ++j) xy[j * K + tt] = fma(f[j], y, xy[j * K + tt]);
            }
        } else if (OP == OP_PEARSON) {
            if (rowok && !shset) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) shx[j] = f[j];
                shset = true;
            }
#pragma unroll
            for (int tt = 0; tt < K; ++tt) {
                {
                    const double y = Yrow[tt];           // shifted by its first valid value on the host
                    const bool ok = rowok && !isnan(y);  // pairwise complete (correlation_test.rs:22-26)
                    // an incomplete pair contributes exact zeros (x = y = 0) instead of being skipped
                    const double ye = ok ? y : 0.0;
                    py[tt] = py[tt] + ye;
                    pyy[tt] = fma(ye, ye, pyy[tt]);
                    pn[tt] = pn[tt] + (ok ? 1.0 : 0.0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const double x = ok ? f[j] - shx[j] : 0.0;
                        const int e = j * K + tt;
                        px[e] = px[e] + x;
                        pxx[e] = fma(x, x, pxx[e]);
                        xy[e] = fma(x, ye, xy[e]);
                    }
                }
            }
        } else if (OP == OP_CHISQ) { // chi2 = total * (sum_j A_j / cs_j - 1), A_j = sum_i f_ij^2 / rowsum_i
            double rsum = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) rsum = rsum + f[j]; // row sum of the frequencies (~1)
            total = total + rsum;
            const double rsd = rowok ? rsum : 1.0;
            const double ri = recip_for_div(rsd);
#pragma unroll
            for (int j = 0; j < NJ; ++j) xx[j] = xx[j] + div_by(f[j] * f[j], rsd, ri);
        }
    }
};

// ---- compact records --------------------------------------------------------------------------------
// What the closing kernel needs of a locus is decided here, where the sums are in registers: which
// alleles survive, in which order the operator uses them, and ONLY the sums of those.  Header (int32):
//   bit 0 alive | bits 1..6 surviving alleles | bit 7 second pass needed | bits 8..10 nk |
//   bits 11..28 ord[0..5], 3 bits each: allele id at rank r
// (ols_iter: stable sort by decreasing column sum, sync.rs:477-506, rank 0 = the major allele that
// ols.rs:227-230 drops; pearson / chisq: surviving alleles in column order).  Doubles, D = nk - 1:
//   ols_iter : per design column d = 0..D-1 (rank d+1):  cs, xy[0..K), xx(d, 0..d)
//   pearson  : py, pyy, pn per trait, then per output d = 0..D-1 (all survivors but the last,
//              correlation_test.rs:95-98): cs, then px, pxx, xy per trait
//   chisq    : chi2 = total * (sum_j A_j / cs_j - 1)   (tables/chisq_test.rs:15-31 regrouped)
// A biallelic ols_iter locus with one trait is 3 doubles + the header instead of 25: the record
// stream is what the HBM pays for twice (written here, read by the closing kernel).
constexpr int H_NK_SHIFT = 8, H_ORD_SHIFT = 11;
template <int K> __device__ __forceinline__ constexpr int ols_field(int d) { return d * (1 + K) + d * (d + 1) / 2; }
template <int K> __device__ __forceinline__ constexpr int prs_field(int d) { return 3 * K + d * (1 + 3 * K); }

// a[idx] for a run-time idx WITHOUT indexing memory: hipcc folds a plain chain of selects over array
// elements back into a dynamically indexed load, which pins the whole accumulator struct in scratch
// memory; the empty asm makes each element an opaque register value first.
__device__ __forceinline__ double opaque(double v) {
    asm volatile("" : "+v"(v));
    return v;
}
template <int N>
__device__ __forceinline__ double pickn(const double (&a)[N], int idx) {
    double r = opaque(a[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) r = (idx == j) ? opaque(a[j]) : r;
    return r;
}
template <int NJ, int K>
__device__ __forceinline__ double pick_trait(const double (&a)[NJ * K], int idx, int t) {
    double r = opaque(a[t]);
#pragma unroll
    for (int j = 1; j < NJ; ++j) r = (idx == j) ? opaque(a[j * K + t]) : r;
    return r;
}

template <int OP, int NJ, int K, typename AJ>
__device__ __forceinline__ void emit_record(const Sums<OP, NJ, K> &S, bool poisoned, const bool (&kp)[NJ], bool alive,
                                            bool again, bool valid, int32_t *__restrict__ rec_flags,
                                            double *__restrict__ rec, int64_t slot, AJ aj, bool sort_desc = false) {
    const size_t rb = rec_base(slot);
    const double pz = poisoned ? NAN : 0.0; // x + NaN = NaN: an uncovered pool makes the plain sums NaN
    int nk = 0, keepmask = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { nk += kp[j] ? 1 : 0; keepmask |= kp[j] ? (2 << aj(j)) : 0; }
    // rank of every surviving slot, and its inverse
    int ordslot[NJ];
    int ordbits = 0;
    {
        int rank[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int r = 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                if (i == j) continue;
                bool before;
                if (OP == OP_OLS || (OP == OP_LOAD && sort_desc))
                    before = S.cs[i] > S.cs[j] || (S.cs[i] == S.cs[j] && i < j);
                else before = i < j;
                r += (kp[i] && before) ? 1 : 0;
            }
            rank[j] = kp[j] ? r : NJ;
        }
#pragma unroll
        for (int r = 0; r < NJ; ++r) {
            int sl = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) sl = (rank[j] == r) ? j : sl;
            ordslot[r] = sl;
            int id = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) id = (sl == j) ? aj(j) : id;
            ordbits |= (r < nk ? id : 0) << (3 * r);
        }
    }
    if (valid)
        rec_flags[slot] = (alive ? FLAG_ALIVE : 0) | keepmask | (again ? FLAG_SECOND : 0) | (nk << H_NK_SHIFT) |
                          (ordbits << H_ORD_SHIFT);
    if (OP == OP_LOAD) return; // the header is all the loader needs
    const int D = (valid && alive && !again) ? nk - 1 : 0; // a locus the second pass redoes gets its sums there
    if (OP == OP_CHISQ) {
        if (D > 0) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc = kp[j] ? acc + S.xx[j] / S.cs[j] : acc;
            rec[rb] = S.total * (acc - 1.0) + pz;
        }
        return;
    }
    if (OP == OP_PEARSON) {
        if (D > 0) {
#pragma unroll
            for (int t = 0; t < K; ++t) {
                rec[rb + (size_t)(3 * t) * 64] = S.py[t];
                rec[rb + (size_t)(3 * t + 1) * 64] = S.pyy[t];
                rec[rb + (size_t)(3 * t + 2) * 64] = S.pn[t];
            }
        }
    }
    static_for<0, NJ - 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if (!__any(d < D)) return; // wave-uniform: nobody has that many columns
        const int a = (OP == OP_OLS) ? ordslot[d + 1] : ordslot[d];
        const bool on = d < D;
        if (OP == OP_OLS) {
            const int f0 = ols_field<K>(d);
            const double c = pickn<NJ>(S.cs, a);
            if (on) rec[rb + (size_t)f0 * 64] = c + pz; // the plain column sum / mean (ols.rs:266, correlation_test.rs:119)
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const double v = pick_trait<NJ, K>(S.xy, a, t);
                if (on) rec[rb + (size_t)(f0 + 1 + t) * 64] = v + pz;
            }
#pragma unroll
            for (int cidx = 0; cidx <= d; ++cidx) {
                const int b2 = ordslot[cidx + 1];
                const int lo = a < b2 ? a : b2, hi = a < b2 ? b2 : a;
                const int ti = lo * NJ - lo * (lo - 1) / 2 + (hi - lo);
                const double v = pickn<NJ * (NJ + 1) / 2>(S.xx, ti);
                if (on) rec[rb + (size_t)(f0 + 1 + K + cidx) * 64] = v + pz;
            }
        } else { // OP_PEARSON
            const int f0 = prs_field<K>(d);
            const double c = pickn<NJ>(S.cs, a);
            if (on) rec[rb + (size_t)f0 * 64] = c + pz; // the plain column sum / mean (ols.rs:266, correlation_test.rs:119)
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const double vx = pick_trait<NJ, K>(S.px, a, t), vxx = pick_trait<NJ, K>(S.pxx, a, t),
                             vxy = pick_trait<NJ, K>(S.xy, a, t);
                if (on) {
                    rec[rb + (size_t)(f0 + 1 + 3 * t) * 64] = vx;
                    rec[rb + (size_t)(f0 + 2 + 3 * t) * 64] = vxx;
                    rec[rb + (size_t)(f0 + 3 + 3 * t) * 64] = vxy;
                }
            }
        }
    });
}

// ---- ols_iter fit from its sufficient statistics (shared by the streaming pass and the closing kernel) ------------
// Literal normal equations in the reference's column order (ols.rs:58-160) on the PN columns [1 | f_ord[1] | ... |
// f_ord[PN-1]] (the major allele, rank 0, is the one ols.rs:227-230 drops), specialised on PN so that the common biallelic
// locus pays for a 2 x 2 factorisation only.  cs = column sums, xxs = sum f_a f_b, xy = sum f_a y_t (y centred on the host).
// Returns `singular` (zero pivot or det(inv) == 0: the locus is dropped, ols.rs:77-83), the diagonal of the inverse through
// the coefficients b and p-values pv of the PN - 1 design columns for the kk traits of this launch.
template <int PN>
__device__ __forceinline__ void ols_solve(const double (&cs)[PN - 1], const double (&xxs)[PN - 1][PN - 1],
                                          const double (&xy)[PN - 1][MAXK], int kk, const LocusParams &P,
                                          const double *__restrict__ tcoef, bool &singular,
                                          double (&bout)[MAXK][PN - 1], double (&pvout)[MAXK][PN - 1]) {
    const int n = P.n;
    double A[PN][PN];
    auto xtx = [&](int r, int c) -> double {
        if (r == 0 && c == 0) return (double)n;
        if (r == 0) return cs[c - 1];
        if (c == 0) return cs[r - 1];
        return xxs[r - 1][c - 1];
    };
#pragma unroll
    for (int r = 0; r < PN; ++r)
#pragma unroll
        for (int c = 0; c < PN; ++c) A[r][c] = xtx(r, c);
    // LU with partial pivoting, first max |a| in the column (the oracle's lu_factor; LAPACK dgetf2)
    singular = false;
    int piv[PN];
#pragma unroll
    for (int kq = 0; kq < PN; ++kq) {
        int pi = kq;
        double pm = fabs(A[kq][kq]);
#pragma unroll
        for (int i2 = kq + 1; i2 < PN; ++i2) {
            const double v = fabs(A[i2][kq]);
            const bool g = v > pm;
            pm = g ? v : pm;
            pi = g ? i2 : pi;
        }
        piv[kq] = pi;
#pragma unroll
        for (int i2 = kq + 1; i2 < PN; ++i2) {
            const bool sw = (pi == i2);
#pragma unroll
            for (int j = 0; j < PN; ++j) {
                const double x1 = A[kq][j], x2 = A[i2][j];
                A[kq][j] = sw ? x2 : x1;
                A[i2][j] = sw ? x1 : x2;
            }
        }
        if (A[kq][kq] == 0.0) singular = true;
        const double inv = 1.0 / A[kq][kq];
#pragma unroll
        for (int i2 = kq + 1; i2 < PN; ++i2) A[i2][kq] = A[i2][kq] * inv;
#pragma unroll
        for (int i2 = kq + 1; i2 < PN; ++i2) {
            const double lf = A[i2][kq];
#pragma unroll
            for (int j = kq + 1; j < PN; ++j) A[i2][j] = A[i2][j] - lf * A[kq][j];
        }
    }
    // x = (X'X)^-1 rhs through the factorisation (P A = L U)
    auto lu_solve = [&](double (&col)[PN]) {
#pragma unroll
        for (int kq = 0; kq < PN; ++kq) {
#pragma unroll
            for (int i2 = kq + 1; i2 < PN; ++i2) {
                const bool sw = (piv[kq] == i2);
                const double x1 = col[kq], x2 = col[i2];
                col[kq] = sw ? x2 : x1;
                col[i2] = sw ? x1 : x2;
            }
        }
#pragma unroll
        for (int i2 = 0; i2 < PN; ++i2) {
            double sacc = col[i2];
#pragma unroll
            for (int j = 0; j < i2; ++j) sacc = sacc - A[i2][j] * col[j];
            col[i2] = sacc;
        }
#pragma unroll
        for (int i2 = PN - 1; i2 >= 0; --i2) {
            double sacc = col[i2];
#pragma unroll
            for (int j = i2 + 1; j < PN; ++j) sacc = sacc - A[i2][j] * col[j];
            col[i2] = sacc / A[i2][i2];
        }
    };
    // full inverse, column by column in the oracle's order: its diagonal gives var(b)
    // (ols.rs:111-116) and its determinant feeds the second singularity test (ols.rs:81-83)
    double Inv[PN][PN], dinv[PN];
#pragma unroll
    for (int c2 = 0; c2 < PN; ++c2) {
        double col[PN];
#pragma unroll
        for (int i2 = 0; i2 < PN; ++i2) col[i2] = (i2 == c2) ? 1.0 : 0.0;
        lu_solve(col);
#pragma unroll
        for (int i2 = 0; i2 < PN; ++i2) Inv[i2][c2] = col[i2];
        dinv[c2] = col[c2];
    }
    // `inv.det() == 0.0` (ols.rs:81): LU of the inverse, singular factorisation -> det 0.
    // Loci with duplicated allele columns pass the first LU by a rounding residue and are
    // caught here, exactly as in the reference.
    {
        double det = 1.0;
        bool zero_piv = false;
#pragma unroll
        for (int kq = 0; kq < PN; ++kq) {
            int pi = kq;
            double pm = fabs(Inv[kq][kq]);
#pragma unroll
            for (int i2 = kq + 1; i2 < PN; ++i2) {
                const double v = fabs(Inv[i2][kq]);
                const bool g = v > pm;
                pm = g ? v : pm;
                pi = g ? i2 : pi;
            }
#pragma unroll
            for (int i2 = kq + 1; i2 < PN; ++i2) {
                const bool sw = (pi == i2);
#pragma unroll
                for (int j = 0; j < PN; ++j) {
                    const double x1 = Inv[kq][j], x2 = Inv[i2][j];
                    Inv[kq][j] = sw ? x2 : x1;
                    Inv[i2][j] = sw ? x1 : x2;
                }
            }
            const double pvt = Inv[kq][kq];
            zero_piv = zero_piv || (pvt == 0.0);
            const double ipv = 1.0 / pvt;
#pragma unroll
            for (int i2 = kq + 1; i2 < PN; ++i2) {
                const double lf = (pvt != 0.0) ? Inv[i2][kq] * ipv : 0.0;
                if (lf != 0.0) {
#pragma unroll
                    for (int j = kq + 1; j < PN; ++j) Inv[i2][j] = Inv[i2][j] - lf * Inv[kq][j];
                }
            }
            det = det * pvt;
        }
        if (zero_piv || det == 0.0) singular = true;
    }
#pragma unroll
    for (int tt = 0; tt < MAXK; ++tt) {
        if (tt >= kk) continue;
        // X'y with the centred phenotype (slopes are invariant to the shift; it removes the
        // y-bar^2 cancellation from the residual sum of squares)
        double xty[PN], b[PN];
        xty[0] = P.sy[tt];
#pragma unroll
        for (int r = 1; r < PN; ++r) xty[r] = xy[r - 1][tt];
#pragma unroll
        for (int r = 0; r < PN; ++r) b[r] = xty[r];
        lu_solve(b);
        // RSS = y'y - 2 b'X'y + b'(X'X) b: the form that is stationary in b, so the O(cond*eps)
        // error of the solve enters only to second order
        double bxy = 0.0;
#pragma unroll
        for (int r = 0; r < PN; ++r) {
            double ab = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < PN; ++c2) ab = fma(xtx(r, c2), b[c2], ab);
            bxy = fma(b[r], 2.0 * xty[r] - ab, bxy);
        }
        double rss = P.syy[tt] - bxy;
        rss = rss < 0.0 ? 0.0 : rss;
        const double ve = rss / ((double)n - (double)PN); // ols.rs:103
#pragma unroll
        for (int r = 0; r < PN - 1; ++r) {
            const double bb = b[r + 1];
            const double vb = ve * dinv[r + 1];                              // ols.rs:111-116
            const double tstat = (fabs(bb) <= PG_EPS) ? 0.0 : bb / sqrt(vb); // ols.rs:143-147
            double pv;
            if (fabs(tstat) <= PG_EPS) pv = 1.0;
            else if (isnan(tstat)) pv = 1.0;
            else pv = pg_t_two_sided_p(fabs(tstat), P.tdf, tcoef, P.ntcoef);
            bout[tt][r] = bb;
            pvout[tt][r] = pv;
        }
    }
}

// pearsons_correlation (gwas/correlation_test.rs:7-71) from the shifted sums over the complete pairs
__device__ __forceinline__ void pearson_close(double sx, double sxx, double sxy, double sy, double syy, double m, int n,
                                              const LocusParams &P, const double *__restrict__ tcoef, double &rr, double &pp) {
    const double cxy = sxy - sx * sy / m;
    const double cxx = sxx - sx * sx / m;
    const double cyy = syy - sy * sy / m;
    const double r0 = cxy / (sqrt(cxx) * sqrt(cyy));      // :50-52
    rr = NAN; pp = NAN;
    if (isnan(r0)) return;                                // :53-56
    const double sden = (1.0 - r0 * r0) / ((double)n - 2.0); // :57
    if (sden <= 0.0) { rr = r0; pp = PG_EPS; return; }    // :58-61
    const double tstat = r0 / sqrt(sden);
    pp = (n > 2) ? pg_t_two_sided_p(fabs(tstat), P.tdf, tcoef, P.ntcoef) : NAN;
    rr = round(r0 * 1e7) / 1e7; // sensible_round(r, 7), :70 (half away from zero)
}

// ---- the streaming pass (round 3): every locus, ONE read of its counts, fits closed in place ------------------------
// One lane per locus, as the filter demands: q_j = sum_i f_ij w_i must be accumulated sequentially in pool order with
// separate multiply and add (sync.rs:258-282), and a lane walking its own row does exactly that.  What changed against
// round 2's k_locus_first (measurements: tools/mb_locus_dma.hip, profiles/r03_mb_locus_*.log):
//  * A lane takes M = period consecutive loci, one after the other, as ONE stream of M * 24 n bytes: that many bytes are a
//    whole number of 128-byte lines (period = 128 / gcd(24 n, 128)), so every lane's stream starts at the same offset inside
//    a line, no line is shared by two lanes (the alignment-class units of round 2 fetched the line two neighbouring rows
//    share twice: +4 % at 100 pools), every lane sees the same pools complete at the same step (the pool loop and the w / Y
//    operands stay wave-uniform), and a unit -- 64 lanes x M loci -- is one contiguous run of loci whose results leave as
//    whole lines.
//  * A stream starts on a pool, so a ring turn of three lines = 384 bytes = 16 pools is unrolled with every LDS offset an
//    immediate (a batch that does not start on a 128-byte boundary is read in windows that straddle two lines: correct,
//    slower); w_i and y_i come from a table in LDS (one broadcast read per pool) instead of scalar loads whose wait also
//    waited for the LDS.  A line of the 64 streams is 8 buffer loads of 16 rows x 64 bytes (the two halves of a row
//    group's line in consecutive instructions), one line ahead of the one being consumed; the descriptor of every line ends
//    with the batch, so the last lines need no clamping.  The pass runs at the rate of this request pattern, which is the
//    same with LDS-DMA or register staging, 2 .. 4 waves per SIMD and any ring depth (tools/mb_locus_dma.hip: 5.4 .. 5.95
//    TB/s by row length): 128 bytes per row and step is the best step (64: 4.8, 32: 3.6, 256: 5.0 TB/s).
//  * Speculation as before: the operator's sums are taken with every candidate allele in play; they ARE the reference's
//    when every allele the filter drops has no reads (the common case).  New: only what a biallelic fit needs is summed
//    (per allele sum f, sum f^2, sum f y -- not the 15 cross products; pearson_corr: ONE regressor z = sum_j j f_j, whose
//    correlation is minus that of the first surviving allele on a biallelic locus; chisq_test: no row-sum division, the
//    total is n - n_missing), and such a fit is CLOSED IN PLACE when its last pool has passed (ols_solve<2> /
//    pearson_close / the chi-square tail) -- no record stream, no closing kernel.  Loci that drop an allele with reads, or
//    (ols_iter, pearson_corr) keep three or more alleles, go to a dense list (one returning atomic per unit that has any)
//    for k_locus_second, which redoes them from the counts and closes them itself.
//  * Results are staged per unit in LDS and leave at the unit's end as contiguous 16-byte non-temporal stores (a unit is a
//    contiguous run of loci): per-lane stores of a locus-strided layout reached HBM as partial lines and cost 130 us per
//    million loci; cached stores another 25 .. 35 us.
//  * Coverage sums are integer sums; a count of 2^29 or more anywhere in the batch raises a flag and the call fails with
//    PG_ERR_INVALID (the sums of 6 x such counts would leave 32 bits) -- the text parser cannot produce one (u32 counts of
//    a sequencing depth).
constexpr int ST_SLOTB = 64 * 128;  // LDS bytes of a wave's line slot: [16 rows x 64 B per load instruction][4 row groups x 2 halves]
constexpr int ST_PPT = 16;          // pools per ring turn of 3 lines
constexpr int ST_STAGE = 8192;      // LDS bytes per wave for the results of a unit

template <int OP, int NJ, int K>
struct Acc { // running sums of one locus; everything accumulated sequentially in pool order
    double q[NJ];         // the filter's pool-size weighted frequencies, every candidate allele in play (sync.rs:258-271), see pool()
    double cs[NJ], dd[NJ]; // LOAD: sum f with every candidate in play -- the reference's when the dropped alleles have no reads (dd: unused)
    // The SPECULATED PAIR (round 4).  The reference recomputes the frequencies on the FILTERED counts (gwas/ols.rs:210-230 ->
    // sync.rs:166-192): one stray read of an allele the filter drops changes every denominator of the locus, so sums taken over all
    // candidates are useless for real data.  Instead the sums of a biallelic fit are taken with the denominator c_a + c_b of a pair
    // chosen on the fly: a = the leading allele of the first covered pool, b = the first OTHER allele that shows a read (the largest
    // of that pool, lowest slot on ties).  Until b shows up, every covered pool has f_a = 1, f_b = 0 whoever b turns out to be, so the
    // choice can wait for the first pool that needs it and nothing is ever redone.  If the filter's survivors are exactly {a, b} the
    // sums ARE the reference's, whatever reads the dropped alleles carry; otherwise (three or more survivors, or an error allele that
    // showed before the real one) the locus is listed for the second pass.
    double cs2[2], dd2[2];   // sum f, sum f^2 of a and b over the pair's own denominators
    double xy2[2 * K];       // OLS: sum f y
    // PEARSON: sums of x = f_a - f_a(first complete pool) over the complete pairs; the allele reported is the LOWER slot of the pair,
    // and corr(f_b, y) = -corr(f_a, y) because f_a + f_b = 1 in every covered pool
    // (sum y, sum y^2 and the number of pairs are the same for every locus whose pools are all covered: LocusParams carries them,
    // formed on the host in this order; a locus with an uncovered pool, or phenotypes with NaN, goes to the second pass)
    double px[K], pxx[K], pxy[K], shx;
    uint32_t mincov, orm;    // smallest coverage of a pool; OR of every count seen
    int n_missing, n_missing2; // pools uncovered over every candidate / over the pair
    int pa, pb;              // slots of a and b, -1 until chosen
    bool shset;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { q[j] = 0.0; cs[j] = 0.0; dd[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { cs2[j] = 0.0; dd2[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) xy2[j] = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) { px[j] = 0.0; pxx[j] = 0.0; pxy[j] = 0.0; }
        shx = 0.0;
        mincov = 0xffffffffu;
        orm = 0u;
        n_missing = 0;
        n_missing2 = 0;
        pa = -1; pb = -1;
        shset = false;
    }
};

struct StreamOut { // where the results go (device pointers of the operator's output arrays)
    int32_t *n_out, *ids;
    double *mf, *stat, *pv;
};

template <int OP, bool RNS, int K>
__global__ __launch_bounds__(LO_THREADS, LO_BLOCKS_DEF) void k_locus_stream(
    const uint32_t *__restrict__ counts, const double *__restrict__ wy, const double *__restrict__ tcoef,
    int32_t *__restrict__ rec_flags, int64_t *__restrict__ second, unsigned long long *__restrict__ second_count,
    const StreamOut O, const LocusParams P, const int M, const int staged) {
    constexpr int NJ = RNS ? 5 : 6;
    constexpr int TW = (OP == OP_OLS || OP == OP_PEARSON) ? 1 + K : 1; // doubles per pool in the table: w_i, y_i0, ...
    constexpr int RECB = 16 + 16 * K;                                  // bytes of a staged result
    auto aj = [](int jj) { return (RNS && jj >= 4) ? jj + 1 : jj; };
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *slotp = lds_raw + wave * ST_SLOTB;
    char *stage = lds_raw + LO_WAVES * ST_SLOTB + wave * ST_STAGE;
    const double *tab = reinterpret_cast<const double *>(lds_raw + LO_WAVES * (ST_SLOTB + ST_STAGE));
    const int n = P.n;
    const int64_t L = P.L;
    {
        double *t = reinterpret_cast<double *>(lds_raw + LO_WAVES * (ST_SLOTB + ST_STAGE));
        for (int i = threadIdx.x; i < TW * n; i += LO_THREADS) t[i] = wy[i];
        __syncthreads();
    }
    const uint32_t rowb = (uint32_t)n * 24u;
    const uint32_t srb = (uint32_t)M * rowb;              // bytes of a lane's stream: a whole number of lines
    const int nh = (int)(srb >> 7);                       // lines per stream
    const uint64_t base0 = reinterpret_cast<uint64_t>(counts);
    const int npool = M * n;                              // pools of a stream
    const int64_t lpu = (int64_t)64 * M;                  // loci per unit
    const int64_t nunits = (L + lpu - 1) / lpu;
    const uint64_t total_bytes = (uint64_t)L * rowb;
    const int64_t wid = (int64_t)blockIdx.x * LO_WAVES + wave, wstride = (int64_t)gridDim.x * LO_WAVES;
    if (wid >= nunits) return;

    // load role of this lane: row rr of a group of 16 rows, 16-byte piece pp of a 64-byte half line
    const int rr = lane & 15, pp = lane >> 4;
    uint32_t vb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) vb[g] = (uint32_t)(16 * g + rr) * srb + (uint32_t)pp * 16u;
    // this lane's own row inside the slot: piece P of the line at ((lane >> 4) * 2 + P / 4) * 1024 + (P % 4) * 256 + (lane & 15) * 16
    const char *cell = slotp + (lane >> 4) * 2048 + (lane & 15) * 16;

    // ---- the request side: one line ahead of the compute side, across the wave's units ------------------------------
    int64_t pre_u = wid;
    int pre_h = 0;
    uint4_t SR[8]; // the line in flight: 8 load instructions of 16 rows x 64 bytes
    auto issue_line = [&]() {
        if (pre_u < nunits) {
            // descriptor of THIS line: base = the unit's start + 128 h, range = what is left of the batch; bytes past the batch come
            // back as zeros (uncovered pools of loci that are never stored)
            const uint64_t uo = (uint64_t)pre_u * 64u * srb + (uint64_t)pre_h * 128u;
            const uint64_t ub = base0 + uo;
            // (rounded up to the 16 bytes of a load: a batch ends on an 8-byte boundary, its base is 16-byte aligned)
            const uint64_t left = total_bytes > uo ? ((total_bytes - uo + 15u) & ~(uint64_t)15) : 0;
            const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)(left > 0xffffffffull ? 0xffffffffu : (uint32_t)left));
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ub >> 32));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char *>(((uint64_t)bhi << 32) | blo), 0, nrec, 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i) // the two halves of a row group's line in consecutive instructions
#ifdef LS_EXP_NTLOAD
                SR[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vb[i >> 1] + (uint32_t)(i & 1) * 64u, 0, 2);
#else
                SR[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vb[i >> 1] + (uint32_t)(i & 1) * 64u, 0, 0);
#endif
            if (++pre_h >= nh) { pre_h = 0; pre_u += wstride; }
        }
    };
    auto land_line = [&]() {
        char *dst = slotp + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4_t *>(dst + i * 1024) = SR[i];
    };

    // ---- per-lane state ------------------------------------------------------------------------------------------------
    Acc<OP, NJ, K> A; // the locus being summed
    // the locus whose last pool has passed, reduced at that moment to what its closing needs (the heavy part runs once per ring
    // turn, after the turn's pools): filter decisions in `fh`, the selected sums in `fv`
    //   OLS     fv = cs, sum f^2, sum f y_t of the design column (the minor allele of a biallelic locus)
    //   PEARSON fv = cs of the reported allele, sum x, sum x^2, sum x y of allele a
    //   CHISQ   fv = cs2[2], dd2[2], total                       LOAD  fv = cs[NJ], cs2[2]
    constexpr int NFV = (OP == OP_OLS) ? 2 + K : (OP == OP_PEARSON) ? 4 : (OP == OP_CHISQ) ? 5 : NJ + 2;
    double fv[NFV];
    int fh = 0;          // bit 0 alive but for the survivor count | 7 band (q to be recomputed) | 11 a pool uncovered over the pair | 12..14 id of the closed allele | 15..20 kept mask by slot | 21 the lower slot of the pair is b | 22..24 slot of a | 25..27 slot of b (7 = none)
    bool fin_pending = false;
    int fin_i = 0;       // which of the lane's M loci it is
    uint32_t orm_unit = 0; // OR of every count this lane has seen
    int64_t cur_u = wid;   // the unit being computed
    int pi = 0, iloc = 0;
    uint2_t cy0 = {0u, 0u}, cy1 = {0u, 0u}; // the last 16 bytes of the previous line

    // the locus in A has seen its last pool: its filter decisions as far as they can be taken here (sync.rs:223-300) and the sums its
    // closing will want; finish() completes the decision (number of survivors, the exact q of a band locus, does the pair hold?)
    auto boundary = [&]() {
        int slotmask = 0;
        bool band = false;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool kpj = !((A.q[j] < P.maf) | (A.q[j] > (1.00 - P.maf)));
            slotmask |= kpj ? (1 << j) : 0;
            band = band || fabs(A.q[j] - P.maf) <= P.qband || fabs(A.q[j] - (1.00 - P.maf)) <= P.qband;
        }
        bool alive0 = !((double)A.mincov < P.min_cov);                                 // sync.rs:227
        alive0 = alive0 && A.n_missing != n;                                           // sync.rs:293
        alive0 = alive0 && !(((double)A.n_missing / (double)n) > P.max_miss);          // sync.rs:297
        // (the counts past the end of the batch read as zeros or as whatever follows the buffer inside its last 16 bytes: the loci they
        // would make up do not exist and must not raise the complaint flag)
        orm_unit |= (cur_u * lpu + (int64_t)lane * M + iloc < L) ? A.orm : 0u;
        const int lo = (A.pa < A.pb) ? 0 : 1;            // which of (a, b) sits in the lower slot
        int idc = 0;
        if constexpr (OP == OP_OLS || OP == OP_PEARSON) {
            int sc = lo; // PEARSON: all surviving alleles but the LAST, unsorted (gwas/correlation_test.rs:94-126): the lower slot of two
            if (OP == OP_OLS) {
                // stable sort by decreasing column sum (sync.rs:477-506): rank 0 = the major allele, dropped (ols.rs:227-230)
                const double ca = lo ? A.cs2[1] : A.cs2[0], cb = lo ? A.cs2[0] : A.cs2[1];
                sc = (cb > ca) ? lo : 1 - lo; // the design column: the minor allele (ties: the lower slot stays first = major)
            }
            const int scslot = sc ? A.pb : A.pa;
#pragma unroll
            for (int j = 0; j < NJ; ++j) idc = (scslot == j) ? aj(j) : idc;
            fv[0] = sc ? A.cs2[1] : A.cs2[0];
            if constexpr (OP == OP_OLS) {
                fv[1] = sc ? A.dd2[1] : A.dd2[0];
#pragma unroll
                for (int t = 0; t < K; ++t) fv[2 + t] = sc ? A.xy2[K + t] : A.xy2[t];
            } else {
                fv[1] = A.px[0]; fv[2] = A.pxx[0]; fv[3] = A.pxy[0];
            }
        } else if constexpr (OP == OP_CHISQ) {
            // chi2 = total * (sum_j A_j / cs_j - 1) with A_j = sum_i f_ij^2 / rowsum_i and total = sum_i rowsum_i (tables/chisq_test.rs:15-31
            // regrouped).  The row sum of a covered pool's frequencies is 1 to an ulp, so A_j = sum f^2 and total = the covered pools,
            // within a few ulp of the reference's value (statistics are compared at 1e-10).  Closed from the pair's sums; three or more
            // survivors go to the second pass
            fv[0] = A.cs2[0]; fv[1] = A.cs2[1]; fv[2] = A.dd2[0]; fv[3] = A.dd2[1];
            fv[4] = (double)(n - A.n_missing2);
        } else {
            // LOAD needs column sums only to order the columns (--keep-p-minus-1): every candidate's when nothing dropped has reads,
            // the pair's for a biallelic locus with stray reads (finish() knows which)
#pragma unroll
            for (int j = 0; j < NJ; ++j) fv[j] = A.cs[j];
            fv[NJ] = A.cs2[0]; fv[NJ + 1] = A.cs2[1];
        }
        fh = (alive0 ? FLAG_ALIVE : 0) | (band ? (1 << 7) : 0) | ((A.n_missing2 > 0) ? (1 << 11) : 0) | (idc << 12) | (slotmask << 15) |
             ((lo != 0) ? (1 << 21) : 0) | ((A.pa & 7) << 22) | ((A.pb & 7) << 25);
    };

    // one pool of this lane's locus
    struct TabRow { double w, y[K]; };
    auto read_tab = [&](int ti) { // broadcast read of the pool's table entry: w_i, y_i
        const double *t = tab + ti * TW;
        TabRow r;
        r.w = t[0];
#pragma unroll
        for (int tt = 0; tt < K; ++tt) r.y[tt] = (TW > 1) ? t[TW > 1 ? 1 + tt : 0] : 0.0;
        return r;
    };
    auto pool = [&](const uint2_t &wa, const uint2_t &wb, const uint2_t &wd, const TabRow &tr, auto chkc) {
        constexpr bool CHK = decltype(chkc)::value;
        const uint32_t c0[6] = {wa.x, wa.y, wb.x, wb.y, wd.x, wd.y};
#ifdef LS_EXP_NOCOMPUTE
        if (c0[0] == 0xdeadbeefu && tr.w == 1.5) A.q[0] += (double)(c0[1] + c0[2] + c0[3] + c0[4] + c0[5]);
        if (CHK) { if (++pi == n) { pi = 0; boundary(); fin_i = iloc++; fin_pending = true; A.clear(); } }
        return;
#endif
        uint32_t ci[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ci[j] = c0[aj(j)];
        // Which slots can hold reads in this pool is a WAVE-UNIFORM question (one scalar branch): on clean biallelic data only A and T
        // do, and the arithmetic of the other slots would add exact zeros to every sum -- skipped (NAC = 2: a third of the work);
        // one stray read in any of the 64 loci of the wave takes the pool through the general code (NAC = NJ).  Same sums either way.
        auto arith = [&](auto nacc) {
            constexpr int NAC = decltype(nacc)::value;
            // coverage of the pool over the alleles in play (sync.rs:217-222 / :170-175), in integers: the reference adds the counts as
            // f64, exact below 2^53, so any order and any exact arithmetic gives its value -- as long as the 32-bit sum does not wrap,
            // which `orm` (every count < 2^29, checked at the end of the unit) guarantees
            uint32_t rsi = ci[0];
            uint32_t orv = ci[0];
#pragma unroll
            for (int j = 1; j < NAC; ++j) { rsi += ci[j]; orv |= ci[j]; }
            A.orm |= orv;
            const bool rowok = rsi != 0u;
            A.mincov = rsi < A.mincov ? rsi : A.mincov;          // sync.rs:223-227
            A.n_missing += rowok ? 0 : 1;
            // an uncovered pool has NaN frequencies in the reference; here they are 0 (all counts are 0, divided by 1) and the pool
            // is counted in n_missing, which poisons / skips what NaN would
            const double rsd = (double)(rsi > 1u ? rsi : 1u);
            const double rinv = recip_for_div(rsd);
            // The filter's q_j = sum_i fl(c_ij / rs_i) * w_i (sync.rs:258-271: sequential, multiply then add) decides which alleles
            // survive and must be decided as the reference decides it -- but its VALUE is needed nowhere else.  So the pass accumulates
            // q~_j = fma(c_ij, fl(w_i * 1/rs_i), q~_j): one operation per allele and pool instead of five, within (n + 8) ulp of q_j
            // (every term is positive).  A locus with some |q~_j - threshold| <= qband (P.qband = 8 (n + 16) eps: never, on real data)
            // has its q recomputed literally in finish(); everybody else's decision is provably the reference's.
            double f[NAC];
            const double wi = tr.w;
#ifdef LS_EXP_QEXACT // (timing experiment: the literal evaluation)
#pragma unroll
            for (int j = 0; j < NAC; ++j) { f[j] = div_by((double)ci[j], rsd, rinv); A.q[j] = A.q[j] + f[j] * wi; }
#else
            {
                const double wr = wi * rinv;
#pragma unroll
                for (int j = 0; j < NAC; ++j) A.q[j] = fma((double)ci[j], wr, A.q[j]);
            }
            if constexpr (OP == OP_LOAD || NAC == 2) {
#pragma unroll
                for (int j = 0; j < NAC; ++j) f[j] = div_by((double)ci[j], rsd, rinv);
            }
#endif
            if constexpr (OP == OP_LOAD) {
#pragma unroll
                for (int j = 0; j < NAC; ++j) A.cs[j] = A.cs[j] + f[j];
            }
            // ---- the speculated pair (see Acc) ------------------------------------------------------------------------------------
            if (A.pb < 0) { // not settled yet: the first pools of a locus, longer where the minor allele is rare
                if (A.pa < 0 && rowok) { // the first covered pool: its leading allele (lowest slot on ties)
                    uint32_t best = ci[0];
                    int bj = 0;
#pragma unroll
                    for (int j = 1; j < NAC; ++j) { const bool g = ci[j] > best; best = g ? ci[j] : best; bj = g ? j : bj; }
                    A.pa = bj;
                }
                if (A.pa >= 0) { // the first OTHER allele that shows a read: the largest of this pool, lowest slot on ties
                    uint32_t best = 0u;
                    int bj = -1;
#pragma unroll
                    for (int j = 0; j < NAC; ++j) {
                        const uint32_t v = (j == A.pa) ? 0u : ci[j];
                        const bool g = v > best;
                        best = g ? v : best;
                        bj = g ? j : bj;
                    }
                    A.pb = bj;
                }
            }
            uint32_t ca = 0u, cb = 0u;
#pragma unroll
            for (int j = 0; j < NAC; ++j) { ca = (A.pa == j) ? ci[j] : ca; cb = (A.pb == j) ? ci[j] : cb; }
            const uint32_t rs2 = ca + cb;
            const bool rowok2 = rs2 != 0u;
            A.n_missing2 += rowok2 ? 0 : 1;
            // frequencies over the pair's own coverage: what to_frequencies gives on the FILTERED counts (sync.rs:166-192) if the pair survives
            double fa, fb;
            if (NAC == 2 && __all(rs2 == rsi)) {
                // every locus of the wave has its pair in {A, T} and nothing else here: the pair's coverage IS the pool's, and its
                // frequencies are the quotients already formed (same operands, same arithmetic: the same bits)
                fa = (A.pa == 0) ? f[0] : ((A.pa == 1) ? f[1] : 0.0);
                fb = (A.pb == 0) ? f[0] : ((A.pb == 1) ? f[1] : 0.0);
            } else {
                const double rsd2 = (double)(rs2 > 1u ? rs2 : 1u);
                const double rinv2 = recip_for_div(rsd2);
                fa = div_by((double)ca, rsd2, rinv2);
                fb = div_by((double)cb, rsd2, rinv2);
            }
            A.cs2[0] = A.cs2[0] + fa;
            A.cs2[1] = A.cs2[1] + fb;
            if constexpr (OP == OP_OLS || OP == OP_CHISQ) {
                A.dd2[0] = fma(fa, fa, A.dd2[0]);
                A.dd2[1] = fma(fb, fb, A.dd2[1]);
            }
            if constexpr (OP == OP_OLS) {
#pragma unroll
                for (int tt = 0; tt < K; ++tt) {
                    const double y = tr.y[tt];
                    A.xy2[tt] = fma(fa, y, A.xy2[tt]);
                    A.xy2[K + tt] = fma(fb, y, A.xy2[K + tt]);
                }
            } else if constexpr (OP == OP_PEARSON) {
                if (rowok2 && !A.shset) { A.shx = fa; A.shset = true; } // shift by the first covered pool's value: small numbers in the sums
#pragma unroll
                for (int tt = 0; tt < K; ++tt) {
                    // (every pair is complete where this locus is closed in place: pools all covered, no NaN phenotype -- boundary())
                    const double y = tr.y[tt];             // shifted by its first valid value on the host
                    const double x = fa - A.shx;
                    A.px[tt] = A.px[tt] + x;
                    A.pxx[tt] = fma(x, x, A.pxx[tt]);
                    A.pxy[tt] = fma(x, y, A.pxy[tt]);
                }
            }
        };
        {
#ifdef LS_EXP_NOFASTPATH
            arith(std::integral_constant<int, NJ>{});
#else
            // (tested for every pool: skipping the test for a while after a pool that needed the general code was measured -- nothing on
            // error-bearing counts, where nearly every pool has a stray read in one of the wave's 64 loci, and -3 % on clean ones)
            uint32_t others = 0u;
#pragma unroll
            for (int j = 2; j < NJ; ++j) others |= ci[j];
            if (__any(others != 0u)) arith(std::integral_constant<int, NJ>{});
            else arith(std::integral_constant<int, 2>{});
#endif
        }
        // the locus' last pool: decide, keep what the closing needs (it runs at the end of the turn) and start the next locus
        if constexpr (CHK) {
            if (++pi == n) {
                pi = 0;
                boundary();
                fin_i = iloc++;
                fin_pending = true;
                A.clear();
            }
        }
    };

    // one staged result: n_out | ids (3 bits each) + nk << 16 | mean frequency | statistic per trait | p-value per trait
    auto put_result = [&](int idx, int nout, int idsp, double mf, const double (&st)[K], const double (&pv)[K]) {
        char *r = stage + (size_t)idx * RECB;
        *reinterpret_cast<uint2_t *>(r) = uint2_t{(uint32_t)nout, (uint32_t)idsp};
        *reinterpret_cast<double *>(r + 8) = mf;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            *reinterpret_cast<double *>(r + 16 + 16 * t) = st[t];
            *reinterpret_cast<double *>(r + 24 + 16 * t) = pv[t];
        }
    };
    // the operator's output arrays of locus l from one result
    auto store_result = [&](int64_t l, int nout, int idsp, double mf, const double (&st)[K], const double (&pv)[K]) {
#ifdef LS_EXP_NOSTORE // (timing experiments only: tools/exp_locus.sh)
        if (nout != 12345) return;
#endif
        const int nk = (idsp >> 16) & 7;
        if (OP == OP_CHISQ) {
            O.n_out[l] = nout;
#pragma unroll
            for (int r = 0; r < PG_MAX_OUT; ++r) O.ids[PG_OIX(l, r)] = (r < nk) ? ((idsp >> (3 * r)) & 7) : -1;
            O.stat[l] = st[0];
            O.pv[l] = pv[0];
            return;
        }
        if (P.t0 == 0 || OP == OP_PEARSON) {
            O.n_out[l] = nout;
#pragma unroll
            for (int r = 0; r < PG_MAX_OUT; ++r) {
                O.ids[PG_OIX(l, r)] = (r < nout) ? (idsp & 7) : -1; // (only single-output loci are closed in this pass)
                O.mf[PG_OIX(l, r)] = (r < nout) ? mf : NAN;
            }
        }
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int r = 0; r < PG_MAX_OUT; ++r) {
                O.stat[PG_OIX(l, r) * P.k_total + P.t0 + t] = (r < nout) ? st[t] : NAN;
                O.pv[PG_OIX(l, r) * P.k_total + P.t0 + t] = (r < nout) ? pv[t] : NAN;
            }
    };

    // ---- a finished locus: close it here or list it for the second pass ------------------------------------------------------------
    auto finish = [&](int64_t unit) {
        const int64_t l = unit * lpu + (int64_t)lane * M + fin_i;
        const bool valid = l < L;
        int slotmask = (fh >> 15) & 63;
        // ---- a locus inside the band: its q as the reference forms it (sync.rs:258-271), from its counts in memory -------------------
        if (__any((fh & (1 << 7)) != 0 && valid)) {
            if ((fh & (1 << 7)) != 0 && valid) {
                double qe[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) qe[j] = 0.0;
                const uint2_t *rp = reinterpret_cast<const uint2_t *>(counts + (size_t)l * (size_t)n * 6);
                for (int i = 0; i < n; ++i) {
                    const uint2_t a0 = rp[3 * i], a1 = rp[3 * i + 1], a2 = rp[3 * i + 2];
                    const uint32_t c6[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
                    uint32_t rs = 0u;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rs += c6[aj(j)];
                    const double rsd = (double)(rs > 1u ? rs : 1u);
                    const double ri = recip_for_div(rsd);
                    const double wi = tab[i * TW];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) qe[j] = qe[j] + div_by((double)c6[aj(j)], rsd, ri) * wi;
                }
                slotmask = 0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) slotmask |= !((qe[j] < P.maf) | (qe[j] > (1.00 - P.maf))) ? (1 << j) : 0;
            }
        }
        int nk = 0, keepmask = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool kpj = (slotmask >> j) & 1;
            nk += kpj ? 1 : 0;
            keepmask |= kpj ? (2 << aj(j)) : 0;
        }
        const bool alive = (fh & FLAG_ALIVE) != 0 && nk >= 2 && valid;                // sync.rs:284 joins the tests of boundary()
        const int pa = (fh >> 22) & 7, pb = (fh >> 25) & 7;
        // did the speculation hold?  the survivors are exactly the pair the sums were taken over
        const bool pair_ok = nk == 2 && pb != 7 && slotmask == ((1 << pa) | (1 << pb));
        const bool pair_poisoned = (fh & (1 << 11)) != 0; // a pool uncovered over the SURVIVORS has NaN frequencies (sync.rs:176-183)
        bool again; // second pass: three or more survivors (cross products, several outputs) or a pair that did not hold
        double pz = pair_poisoned ? NAN : 0.0; // x + NaN = NaN: an uncovered pool makes the reference's plain sums NaN
        bool load_clean = true;
        if constexpr (OP == OP_OLS || OP == OP_CHISQ) again = alive && !pair_ok;
        else if constexpr (OP == OP_PEARSON) // an incomplete pair (uncovered pool, NaN phenotype) changes sum y, sum y^2 and the pair count
            again = alive && (!pair_ok || pair_poisoned || P.y_complete == 0);
        else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) // "has reads" = a positive column sum of frequencies (non-negative terms do not cancel)
                load_clean = load_clean && (((slotmask >> j) & 1) || fv[j < NFV ? j : 0] == 0.0);
            again = alive && P.sort_desc != 0 && !load_clean && !pair_ok;
        }
        const bool deferred = again;
        if (OP == OP_LOAD) {
            // the loader wants the header of every locus: surviving alleles in the order its columns take (sync.rs:1033-1037)
            double key[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                key[j] = load_clean ? fv[j < NFV ? j : 0] : (j == pa ? fv[NJ < NFV ? NJ : 0] : (j == pb ? fv[(NJ + 1) < NFV ? NJ + 1 : 0] : 0.0));
            int ordbits = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int r = 0;
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    if (i == j) continue;
                    const bool before = P.sort_desc ? (key[i] > key[j] || (key[i] == key[j] && i < j)) : i < j;
                    r += (((slotmask >> i) & 1) && before) ? 1 : 0;
                }
                ordbits |= ((slotmask >> j) & 1) ? (aj(j) << (3 * r)) : 0;
            }
            const int hdr = (alive ? FLAG_ALIVE : 0) | keepmask | (again ? FLAG_SECOND : 0) | (nk << H_NK_SHIFT) | (ordbits << H_ORD_SHIFT);
            if (staged & 1) *reinterpret_cast<int32_t *>(stage + (size_t)(lane * M + fin_i) * 4) = hdr;
            else if (valid) rec_flags[l] = hdr;
        } else if (deferred) {
            rec_flags[l] = FLAG_ALIVE | keepmask | FLAG_SECOND | (nk << H_NK_SHIFT); // the second pass wants the surviving alleles
        }
        // the list of the second pass: one returning atomic per unit that has any (its wait drains the line in flight)
        {
            const unsigned long long bal = __ballot(deferred);
            if (bal) {
                unsigned long long basev = 0;
                if (lane == 0) basev = atomicAdd(second_count + SC_LIST, (unsigned long long)__popcll(bal));
                basev = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(basev >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)basev);
                if (deferred) second[basev + __popcll(bal & ((1ull << lane) - 1ull))] = l;
            }
        }
        if (OP == OP_LOAD) return;
        // ---- close the locus ------------------------------------------------------------------------------------------------------
        int nout = 0, idsp = 0;
        double mf = NAN, st[K], pv[K];
#pragma unroll
        for (int t = 0; t < K; ++t) { st[t] = NAN; pv[t] = NAN; }
#ifdef LS_EXP_NOCLOSE
        const bool simple = alive && !deferred && nk == 12345;
#else
        const bool simple = alive && !deferred;
#endif
        if (OP == OP_CHISQ) {
            // every surviving allele is listed (column order), alive or not (the row of the locus)
            int r = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool kpj = (slotmask >> j) & 1;
                idsp |= kpj ? (aj(j) << (3 * r)) : 0;
                r += kpj ? 1 : 0;
            }
            idsp |= nk << 16;
            if (__any(simple)) {
                const double acc = fv[2 < NFV ? 2 : 0] / fv[0] + fv[3 < NFV ? 3 : 0] / fv[1 < NFV ? 1 : 0];
                const double chi2 = fv[4 < NFV ? 4 : 0] * (acc - 1.0) + pz; // tables/chisq_test.rs:15-31 regrouped
                const double df = (double)(n * nk) - 1.0;
                const double p = pg_chisq_upper_p(chi2, df, pg_ln_gamma(df / 2.0)); // :33-35
                if (simple) { st[0] = chi2; pv[0] = p; nout = nk; }
            }
        } else if (__any(simple)) {
            if (OP == OP_OLS) {
                double cs1[1], xx1[1][1], xy1[1][MAXK], b1[MAXK][1], p1[MAXK][1];
                cs1[0] = fv[0] + pz;
                xx1[0][0] = fv[1 < NFV ? 1 : 0] + pz;
#pragma unroll
                for (int t = 0; t < MAXK; ++t) xy1[0][t] = (t < K) ? fv[(2 + t) < NFV ? 2 + t : 0] + pz : 0.0;
                bool singular;
                ols_solve<2>(cs1, xx1, xy1, K, P, tcoef, singular, b1, p1);
                if (simple && !singular) { // Err -> the whole locus is dropped (ols.rs:250-253)
                    nout = 1;
                    idsp = (fh >> 12) & 7;
                    mf = cs1[0] / (double)n; // ols.rs:266
#pragma unroll
                    for (int t = 0; t < K; ++t) { st[t] = b1[t][0]; pv[t] = p1[t][0]; }
                }
            } else { // OP_PEARSON
                double rr, pp;
                pearson_close(fv[1 < NFV ? 1 : 0], fv[2 < NFV ? 2 : 0], fv[3 < NFV ? 3 : 0], P.py[0], P.pyy[0], P.pn[0], n, P, tcoef, rr, pp);
                if (simple) {
                    nout = 1;
                    idsp = (fh >> 12) & 7;
                    mf = (fv[0] + pz) / (double)n; // x.mean(), :119: the plain mean is NaN with an uncovered pool
                    st[0] = (fh & (1 << 21)) ? -rr : rr; // the sums are those of f_a; reported is the lower slot: if that is b, f_b = 1 - f_a changes the sign
                    pv[0] = pp;
                }
            }
        }
        // a deferred locus gets the "dropped" pattern here; k_locus_second overwrites it later in the stream
        if (staged & 1) put_result(lane * M + fin_i, nout, idsp, mf, st, pv);
        else if (valid) store_result(l, nout, idsp, mf, st, pv);
    };

    // the unit's results leave as contiguous runs: the loci of a unit are consecutive, so every output array has ONE contiguous
    // region per unit, written in 16-byte pieces by consecutive lanes (piece -> its elements -> (locus, slot) -> the staged result).
    // Per-lane stores of a locus' own 4 .. 8 bytes would reach memory as partial lines (measured with the former locus-major layout:
    // 87 us per million loci at 100 pools, a fifth of the pass).
    const bool coalesced = (staged & 2) != 0; // the host checked: 16-byte aligned arrays, every trait of the call in this launch
    auto flush_unit = [&](int64_t unit) {
        __builtin_amdgcn_wave_barrier();
#ifdef LS_EXP_SMALLOUT // (timing experiment: every unit writes the first unit's region -- the stores stay, the memory traffic goes)
        const int64_t l0 = 0;
        const int nv = (int)lpu;
#else
        const int64_t l0 = unit * lpu;
        const int nv = (int)((L - l0) < lpu ? (L - l0) : lpu); // loci of this unit that exist
#endif
        if (OP == OP_LOAD) {
            for (int idx = lane; idx < nv; idx += 64) rec_flags[l0 + idx] = *reinterpret_cast<const int32_t *>(stage + (size_t)idx * 4);
        } else if (!coalesced) {
            for (int idx = lane; idx < nv; idx += 64) {
                const char *rp = stage + (size_t)idx * RECB;
                const uint2_t h = *reinterpret_cast<const uint2_t *>(rp);
                double st[K], pv[K];
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    st[t] = *reinterpret_cast<const double *>(rp + 16 + 16 * t);
                    pv[t] = *reinterpret_cast<const double *>(rp + 24 + 16 * t);
                }
                store_result(l0 + idx, (int)h.x, (int)h.y, *reinterpret_cast<const double *>(rp + 8), st, pv);
            }
        } else {
#ifndef LS_EXP_NOSTORE
            // one array: EPL elements per locus, element (j, sub) = get(j, sub); 16 bytes per lane and store
            auto emit32 = [&](int32_t *base, auto eplc, auto get) {
                constexpr int EPL = decltype(eplc)::value;
                const int ne = nv * EPL;
                int32_t *dst = base + l0 * EPL;
                if (reinterpret_cast<uintptr_t>(dst) & 15) { // a slot array whose start is not on 16 bytes (L not a multiple of 4)
                    for (int e = lane; e < ne; e += 64) { const int j = e / EPL; dst[e] = get(j, e - j * EPL); }
                    return;
                }
                for (int c = lane; c * 4 < ne; c += 64) {
                    int v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = c * 4 + u;
                        const int j = e / EPL;
                        v[u] = get(j < nv ? j : nv - 1, e - j * EPL);
                    }
#ifndef LS_EXP_PLAINSTORE // results are written once and not read by this launch: non-temporal (measured: -25 .. -35 us per million loci)
                    if (c * 4 + 3 < ne) __builtin_nontemporal_store(uint4_t{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]}, reinterpret_cast<uint4_t *>(dst + c * 4));
#else
                    if (c * 4 + 3 < ne) *reinterpret_cast<uint4_t *>(dst + c * 4) = uint4_t{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
#endif
                    else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (c * 4 + u < ne) dst[c * 4 + u] = v[u];
                    }
                }
            };
            auto emit64 = [&](double *base, auto eplc, auto get) {
                constexpr int EPL = decltype(eplc)::value;
                const int ne = nv * EPL;
                double *dst = base + l0 * EPL;
                if (reinterpret_cast<uintptr_t>(dst) & 15) {
                    for (int e = lane; e < ne; e += 64) { const int j = e / EPL; dst[e] = get(j, e - j * EPL); }
                    return;
                }
                for (int c = lane; c * 2 < ne; c += 64) {
                    double v[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = c * 2 + u;
                        const int j = e / EPL;
                        v[u] = get(j < nv ? j : nv - 1, e - j * EPL);
                    }
#ifndef LS_EXP_PLAINSTORE
                    typedef double d2v __attribute__((ext_vector_type(2)));
                    if (c * 2 + 1 < ne) __builtin_nontemporal_store(d2v{v[0], v[1]}, reinterpret_cast<d2v *>(dst + c * 2));
#else
                    if (c * 2 + 1 < ne) *reinterpret_cast<double2 *>(dst + c * 2) = double2{v[0], v[1]};
#endif
                    else dst[c * 2] = v[0];
                }
            };
            auto hdr_of = [&](int j) { return *reinterpret_cast<const uint2_t *>(stage + (size_t)j * RECB); };
            auto dbl_of = [&](int j, int off) { return *reinterpret_cast<const double *>(stage + (size_t)j * RECB + off); };
            if (OP == OP_CHISQ) {
                emit32(O.n_out, std::integral_constant<int, 1>{}, [&](int j, int) { return (int)hdr_of(j).x; });
                // the surviving alleles of the row: the slots some locus of the unit needs (slots r >= n_out[l] are unspecified)
                int mx = 0;
                for (int idx = lane; idx < nv; idx += 64) mx = max(mx, (int)((hdr_of(idx).y >> 16) & 7u));
                for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
                mx = __builtin_amdgcn_readfirstlane(mx);
#pragma unroll
                for (int r = 0; r < PG_MAX_OUT; ++r)
                    if (r < mx)
                        emit32(O.ids + (size_t)r * P.L, std::integral_constant<int, 1>{}, [&](int j, int) {
                            const int idsp = (int)hdr_of(j).y;
                            return r < ((idsp >> 16) & 7) ? ((idsp >> (3 * r)) & 7) : -1;
                        });
                emit64(O.stat, std::integral_constant<int, 1>{}, [&](int j, int) { return dbl_of(j, 16); });
                emit64(O.pv, std::integral_constant<int, 1>{}, [&](int j, int) { return dbl_of(j, 24); });
            } else {
                // (only single-output loci are closed in this pass: slot 0 of every array, the other slots are not written)
                if (P.t0 == 0 || OP == OP_PEARSON) {
                    emit32(O.n_out, std::integral_constant<int, 1>{}, [&](int j, int) { return (int)hdr_of(j).x; });
                    emit32(O.ids, std::integral_constant<int, 1>{}, [&](int j, int) {
                        const uint2_t h = hdr_of(j);
                        return h.x ? (int)(h.y & 7u) : -1;
                    });
                    emit64(O.mf, std::integral_constant<int, 1>{}, [&](int j, int) { return hdr_of(j).x ? dbl_of(j, 8) : NAN; });
                }
                // [slot 0][locus][trait], every trait of the call in this launch (k_total == K, t0 == 0)
                emit64(O.stat, std::integral_constant<int, K>{}, [&](int j, int t) { return hdr_of(j).x ? dbl_of(j, 16 + 16 * t) : NAN; });
                emit64(O.pv, std::integral_constant<int, K>{}, [&](int j, int t) { return hdr_of(j).x ? dbl_of(j, 24 + 16 * t) : NAN; });
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();
    };

    if (n < ST_PPT) {
        // fewer pools than a ring turn holds: several loci could end inside one turn.  Such rows are at most 360 bytes; every lane
        // reads its own stream straight from memory, pool by pool (same sums, same closing, no staging of lines)
        for (; cur_u < nunits; cur_u += wstride) {
            A.clear();
            pi = 0; iloc = 0; fin_pending = false;
            const uint64_t so = ((uint64_t)cur_u * 64u + (uint64_t)lane) * srb; // this lane's stream
            for (int sp = 0; sp < npool; ++sp) {
                const uint64_t o = so + (uint64_t)sp * 24u;
                uint2_t wv[3] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
                if (o + 24u <= total_bytes) {
                    const uint2_t *g = reinterpret_cast<const uint2_t *>(reinterpret_cast<const char *>(counts) + o);
                    wv[0] = g[0]; wv[1] = g[1]; wv[2] = g[2];
                }
                pool(wv[0], wv[1], wv[2], read_tab(pi), std::true_type{});
                if (fin_pending) {
                    finish(cur_u);
                    fin_pending = false;
                }
            }
            if (staged & 1) flush_unit(cur_u);
            if (__any((orm_unit >> 29) != 0u)) {
                if (lane == 0) atomicOr(second_count + SC_COMPLAINT, 1ull);
                orm_unit = 0;
            }
        }
        return;
    }
    // ---- one step: land the line that was in flight, request the next one, run the pools that end in this line ----------
    issue_line();
    while (cur_u < nunits) {
        A.clear();
        pi = 0; iloc = 0; fin_pending = false;
        const int nturn = nh / 3; // a stream is a whole number of lines AND of pools: a whole number of turns (384 = lcm(128, 24))
        // a ring turn: three lines, sixteen pools
        auto turn = [&](auto chkc) {
            static_for<0, 3>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                __builtin_amdgcn_wave_barrier();
                land_line(); // the compiler waits for the registers
                __builtin_amdgcn_wave_barrier();
                issue_line(); // in flight while this line is computed
                // the pools whose LAST byte lies in this line: slots JFIRST .. JLAST of the turn
                constexpr int JFIRST = (s * 128) / 24;            // first j with 24 j + 23 >= 128 s
                constexpr int JLAST = ((s + 1) * 128 - 24) / 24;  // last j with 24 j + 23 < 128 (s + 1)
                auto read_words = [&](auto jc, uint2_t (&wv)[3]) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int uw = 24 * j + 8 * k - s * 128; // offset inside this line; < 0: one of the two carried words
                        if (uw >= 0) wv[k] = *reinterpret_cast<const uint2_t *>(cell + (uw / 64) * 1024 + ((uw / 16) & 3) * 256 + (uw & 8));
                        else wv[k] = (uw == -16) ? cy0 : cy1;
                    }
                };
                // one pool ahead, by hand: the reads of pool j + 1 are issued in front of the arithmetic of pool j (the test for the
                // locus' end makes every pool a basic block of its own, and the compiler does not move loads across those: left alone,
                // every pool exposed two LDS round trips)
                uint2_t wc[3], wn[3];
                TabRow tc, tn;
                read_words(std::integral_constant<int, JFIRST>{}, wc);
                tc = read_tab(pi);
                static_for<JFIRST, JLAST + 1>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j < JLAST) {
                        read_words(std::integral_constant<int, j + 1>{}, wn);
                        tn = read_tab(pi + 1 == n ? 0 : pi + 1);
                    }
                    pool(wc[0], wc[1], wc[2], tc, chkc);
                    if constexpr (j < JLAST) {
                        wc[0] = wn[0]; wc[1] = wn[1]; wc[2] = wn[2];
                        tc = tn;
                    }
                });
                { // the last 16 bytes of this line, for a pool that starts here and ends in the next line
                    const uint4_t tl = *reinterpret_cast<const uint4_t *>(cell + 1024 + 3 * 256);
                    cy0 = uint2_t{tl.x, tl.y};
                    cy1 = uint2_t{tl.z, tl.w};
                }
            });
        };
        for (int T = 0; T < nturn; ++T) {
            turn(std::true_type{});
            if (fin_pending) { // (at most one locus ends per turn: n >= 16 pools, checked by the host)
#ifndef LS_EXP_NOFINISH
                finish(cur_u);
#endif
                fin_pending = false;
            }
        }
#ifndef LS_EXP_NOFLUSH
        if (staged & 1) flush_unit(cur_u);
#endif
        if (__any((orm_unit >> 29) != 0u)) { // a count the 32-bit coverage sums cannot take: the host reports it
            if (lane == 0) atomicOr(second_count + SC_COMPLAINT, 1ull);
            orm_unit = 0;
        }
        cur_u += wstride;
    }
}

// ---- closing arithmetic of a locus of the second pass' list, from the compact record (see emit_record) -----
template <int PN>
__device__ __forceinline__ void ols_close(const double *rec, size_t rb, int k, int ordbits, bool alive,
                                          const double *__restrict__ tcoef, int32_t *__restrict__ n_out,
                                          int32_t *__restrict__ ids_out, double *__restrict__ mf_out,
                                          double *__restrict__ stat_out, double *__restrict__ pv_out, int64_t l,
                                          const LocusParams &P) {
    constexpr int D = PN - 1;
    const int n = P.n;
    const int kk = k; // 1 or 2 traits in this launch: the record was laid out with K = k
    double cs[D], xxs[D][D], xy[D][MAXK];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int f0 = d * (1 + kk) + d * (d + 1) / 2;
        cs[d] = rec[rb + (size_t)f0 * 64];
#pragma unroll
        for (int t = 0; t < MAXK; ++t) xy[d][t] = (t < kk) ? rec[rb + (size_t)(f0 + 1 + t) * 64] : 0.0;
#pragma unroll
        for (int c = 0; c <= d; ++c) {
            const double v = rec[rb + (size_t)(f0 + 1 + kk + c) * 64];
            xxs[d][c] = v;
            xxs[c][d] = v;
        }
    }
    bool singular;
    double bo[MAXK][D], po[MAXK][D];
    ols_solve<PN>(cs, xxs, xy, kk, P, tcoef, singular, bo, po);
    const bool ok = alive && !singular; // Err -> the whole locus is dropped (ols.rs:250-253)
    if (P.t0 == 0) {
        n_out[l] = ok ? D : 0;
#pragma unroll
        for (int r = 0; r < PG_MAX_OUT; ++r) {
            const bool on = ok && r < D;
            ids_out[PG_OIX(l, r)] = on ? ((ordbits >> (3 * (r + 1))) & 7) : -1;
            mf_out[PG_OIX(l, r)] = on ? cs[r < D ? r : 0] / (double)n : NAN; // ols.rs:266
        }
    }
#pragma unroll
    for (int tt = 0; tt < MAXK; ++tt) {
        if (tt >= k) continue;
#pragma unroll
        for (int r = 0; r < PG_MAX_OUT; ++r) {
            const bool on = ok && r < D;
            stat_out[PG_OIX(l, r) * P.k_total + P.t0 + tt] = on ? bo[tt][r < D ? r : 0] : NAN;
            pv_out[PG_OIX(l, r) * P.k_total + P.t0 + tt] = on ? po[tt][r < D ? r : 0] : NAN;
        }
    }
}

template <int OP, int PNFIX = 0>
__device__ __forceinline__ void close_locus(const int64_t l, const int32_t *rec_flags,
                                            const double *rec, const double *__restrict__ tcoef,
                                            int32_t *__restrict__ n_out, int32_t *__restrict__ ids_out,
                                            double *__restrict__ mf_out, double *__restrict__ stat_out,
                                            double *__restrict__ pv_out, const LocusParams &P) {
    // (only the loci the streaming pass listed come here: everything else was closed there)
    do {
        const int n = P.n, k = P.k;
        const int64_t slot = unit_slot(l, P.pshift);
        const size_t rb = rec_base(slot);
        const int hdr = rec_flags[slot];
        const bool alive = (hdr & FLAG_ALIVE) != 0;
        const int nk = (hdr >> H_NK_SHIFT) & 7;
        const int ordbits = hdr >> H_ORD_SHIFT;

        if (OP == OP_CHISQ) {
            // tables/chisq_test.rs:15-35 on the frequency table of the surviving alleles
#pragma unroll
            for (int r = 0; r < PG_MAX_OUT; ++r) ids_out[PG_OIX(l, r)] = (r < nk) ? ((ordbits >> (3 * r)) & 7) : -1;
            const double chi2 = alive ? rec[rb] : NAN;
            const double df = (double)(n * nk) - 1.0;
            n_out[l] = alive ? nk : 0;
            stat_out[l] = chi2;
            pv_out[l] = alive ? pg_chisq_upper_p(chi2, df, pg_ln_gamma(df / 2.0)) : NAN;
            break;
        }

        if (OP == OP_PEARSON) {
            // gwas/correlation_test.rs:94-126: all surviving alleles but the LAST, unsorted
            const int nout = alive ? (nk >= 2 ? nk - 1 : nk) : 0;
            n_out[l] = nout;
#pragma unroll
            for (int r = 0; r < PG_MAX_OUT; ++r) {
                const bool on = r < nout;
                const int f0 = 3 * k + r * (1 + 3 * k);
                ids_out[PG_OIX(l, r)] = on ? ((ordbits >> (3 * r)) & 7) : -1;
                mf_out[PG_OIX(l, r)] = on ? rec[rb + (size_t)f0 * 64] / (double)n : NAN; // x.mean(), :119
#pragma unroll
                for (int tt = 0; tt < MAXK; ++tt) {
                    if (tt >= k) continue;
                    double rr = NAN, pp = NAN;
                    if (on)
                        pearson_close(rec[rb + (size_t)(f0 + 1 + 3 * tt) * 64], rec[rb + (size_t)(f0 + 2 + 3 * tt) * 64],
                                      rec[rb + (size_t)(f0 + 3 + 3 * tt) * 64], rec[rb + (size_t)(3 * tt) * 64],
                                      rec[rb + (size_t)(3 * tt + 1) * 64], rec[rb + (size_t)(3 * tt + 2) * 64], n, P, tcoef, rr, pp);
                    stat_out[PG_OIX(l, r) * P.k_total + P.t0 + tt] = rr;
                    pv_out[PG_OIX(l, r) * P.k_total + P.t0 + tt] = pp;
                }
            }
            break;
        }

        // ---------------- OP_OLS ----------------------------------------------------------------------
        const int pn = alive ? nk : 0;
        if (pn < 2) { // not emitted: ols.rs:215-237
            if (P.t0 == 0) {
                n_out[l] = 0;
#pragma unroll
                for (int r = 0; r < PG_MAX_OUT; ++r) { ids_out[PG_OIX(l, r)] = -1; mf_out[PG_OIX(l, r)] = NAN; }
            }
            for (int tt = 0; tt < k; ++tt)
#pragma unroll
                for (int r = 0; r < PG_MAX_OUT; ++r) {
                    stat_out[PG_OIX(l, r) * P.k_total + P.t0 + tt] = NAN;
                    pv_out[PG_OIX(l, r) * P.k_total + P.t0 + tt] = NAN;
                }
        }
        static_for<2, NA + 1>([&](auto pc) {
            constexpr int PNc = decltype(pc)::value;
            if constexpr (PNFIX == 0 || PNFIX == PNc) { // (a tile of the second pass holds loci with ONE number of survivors)
                if (pn == PNc)
                    ols_close<PNc>(rec, rb, k, ordbits, alive, tcoef, n_out, ids_out, mf_out, stat_out, pv_out, l, P);
            }
        });
    } while (false);
}

// ---- ols_iter and chisq_test, order-free (round 4): a LOCUS PER ROW OF LANES instead of a locus per lane ----------------------------
// The streaming pass above reads with the lane-per-locus request pattern (0.68 - 0.74 of the HBM peak with nothing else going on) because
// the filter's q must be summed in pool order -- and it has to SPECULATE on the surviving pair, because it sees a locus once.  ols_iter
// and chisq_test do not need pool-order sums: their decisions (which alleles survive, which is the major one, is the fit singular)
// must be the reference's, but their NUMBERS -- beta, chi2, p, the mean frequency ols_iter prints with 8 decimals -- only to 1e-10.
// So here a wave reads whole loci with plain coalesced 16-byte loads into a wave-private LDS buffer (a group of 64 / LPL loci,
// <= 10.5 KB; the next group waits in registers meanwhile), LPL = 16 / 32 / 64 lanes share one locus (pool = round * LPL + lane), and
// the sums are reduced over the lanes of the row in a fixed butterfly order (the result of a locus does not depend on its place):
//   phase 1  q~_j of every candidate allele in SINGLE precision, coverage minimum, missing pools -> the filter's decisions; a locus
//            with some q~ within 2e-4 (relative) of a threshold has its q recomputed literally (fp64, pool order, multiply then add);
//   phase 2  the buffer is read again: frequencies over the SURVIVORS' coverage (the reference recomputes them on the filtered
//            counts, gwas/ols.rs:210-230 -> sync.rs:166-192).  ols_iter, two survivors: sum f, sum f^2, sum f y of the rarer one (the
//            other's follow from f_a + f_b = 1); chisq_test: any number of survivors.
// No speculation -- the survivors are known before the sums are taken -- so stray reads of dropped alleles cost nothing, and the
// pass runs at the same 0.62 of the HBM peak (100 pools) whatever the counts look like; on clean counts the streaming pass is
// faster (0.68 - 0.70), which is why a context switches between the two (launch_passes).  What stays with the exact second pass
// (k_locus_second, pool-order sums): ols_iter loci with three or more survivors, and those whose decision could depend on the order
// of the sums -- column sums of the two survivors within 1e-9 n of each other (which one is the major allele), a design within
// 1e-8 of singular (ols.rs:77-83).  pearson_corr prints a full-precision mean and keeps the streaming pass (built here too and
// measured: the pool-order sum of the mean as an n-step chain on the row's first lane, frequencies handed over in LDS, parity
// green -- 0.62 ms against 0.59 for streaming pass + second pass on the same box; not kept).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
// all LPL lanes of a locus end up with the same total, combined in the same order (xor 1, 2, mirror in 8, mirror in 16, xor 16, xor 32)
template <int LPL, typename T, typename OPF>
__device__ __forceinline__ T row_all(T v, OPF op) {
    if constexpr (std::is_same<T, double>::value) {
        v = op(v, dpp_d<0xB1>(v)); v = op(v, dpp_d<0x4E>(v)); v = op(v, dpp_d<0x141>(v)); v = op(v, dpp_d<0x140>(v));
    } else {
        v = op(v, (T)dpp_i<0xB1>((int)v)); v = op(v, (T)dpp_i<0x4E>((int)v)); v = op(v, (T)dpp_i<0x141>((int)v)); v = op(v, (T)dpp_i<0x140>((int)v));
    }
    if constexpr (LPL >= 32) v = op(v, __shfl_xor(v, 16));
    if constexpr (LPL >= 64) v = op(v, __shfl_xor(v, 32));
    return v;
}
#ifndef RW_UNROLL
#define RW_UNROLL 2
#endif
constexpr int RW_MAXR = 7;              // rounds per locus at most (112 / 16, 224 / 32, 448 / 64 pools per lane)
constexpr int RW_NP = 11;               // 16-byte pieces per lane and group: 11 KB >= 24 bytes x 448 pools (x 2 loci x 224, x 4 x 112)
constexpr int RW_BUF = RW_NP * 1024;
#ifndef RW_DIRECT_OCC
#define RW_DIRECT_OCC 3
#endif
#ifndef RW_DIRECT_SETS
#define RW_DIRECT_SETS 1
#endif

template <int OP, int LPL, bool RNS, int K, bool DIRECT>
__global__ __launch_bounds__(LO_THREADS, (DIRECT ? RW_DIRECT_OCC : 2)) void k_ols_rows(
    const uint32_t *__restrict__ counts, const double *__restrict__ wy, const double *__restrict__ tcoef,
    int32_t *__restrict__ rec_flags, int64_t *__restrict__ second, unsigned long long *__restrict__ second_count,
    const StreamOut O, const LocusParams P, const int coalesced) {
    constexpr int NJ = RNS ? 5 : 6;
    constexpr int GL = 64 / LPL;            // loci per group
    constexpr int TW = (OP == OP_OLS) ? 1 + K : 1; // doubles per pool in the table: w_i (, y_i0, ...)
    constexpr int RECB = 16 + 16 * K;
    constexpr int NSUM = 2 + K;             // staged per locus: cs, sum f^2, sum f y_t of the design column (chisq_test: sum_j A_j / cs_j, covered pools)
    auto aj = [](int jj) { return (RNS && jj >= 4) ? jj + 1 : jj; };
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int BUFB = DIRECT ? 0 : RW_BUF; // DIRECT: no staging buffer -- a lane loads its own pools (see RowRegs)
    constexpr int PER_WAVE = BUFB + 64 * NSUM * 8 + 64 * 4 + 64 * RECB;
    char *buf = lds_raw + wave * PER_WAVE;
    double *sums = reinterpret_cast<double *>(buf + BUFB);
    int32_t *hdrs = reinterpret_cast<int32_t *>(buf + BUFB + 64 * NSUM * 8);
    char *stage = buf + BUFB + 64 * NSUM * 8 + 64 * 4;
    const double *tab = reinterpret_cast<const double *>(lds_raw + LO_WAVES * PER_WAVE);
    const int n = P.n;
    const int64_t L = P.L;
    const float *tabf = reinterpret_cast<const float *>(lds_raw + LO_WAVES * PER_WAVE + sizeof(double) * TW * n); // the weights in fp32
    {
        double *t = reinterpret_cast<double *>(lds_raw + LO_WAVES * PER_WAVE);
        float *tf = reinterpret_cast<float *>(lds_raw + LO_WAVES * PER_WAVE + sizeof(double) * TW * n);
        for (int i = threadIdx.x; i < TW * n; i += LO_THREADS) t[i] = wy[i];
        for (int i = threadIdx.x; i < n; i += LO_THREADS) tf[i] = (float)wy[i * TW];
        __syncthreads();
    }
    const uint32_t rowb = (uint32_t)n * 24u;
    const uint32_t grpb = (uint32_t)GL * rowb;                 // bytes of a group: a multiple of 16 (GL n even: the host checked)
    const uint64_t total_bytes = (uint64_t)L * rowb;
    const uint64_t base0 = reinterpret_cast<uint64_t>(counts);
    const int64_t nunits = (L + 63) / 64;
    const int64_t wid = (int64_t)blockIdx.x * LO_WAVES + wave, wstride = (int64_t)gridDim.x * LO_WAVES;
    if (wid >= nunits) return;
    const int row = lane / LPL, li = lane % LPL;
    const char *lbase = buf + (uint32_t)row * rowb;

    // ---- the request side: the next group's 16-byte pieces wait in registers while the current group is summed -----------------------
    uint4_t SR[RW_NP];
    int64_t pre_u = wid;
    int pre_g = 0;
    auto issue_group = [&]() {
        const uint64_t go = ((uint64_t)pre_u * 64u + (uint64_t)pre_g * GL) * rowb; // (past the batch: the descriptor answers with zeros)
        const uint64_t left = total_bytes > go ? total_bytes - go : 0;
        const uint64_t lim = left < grpb ? left : grpb;                          // this group's bytes only: the rest of the buffer stays zero
        const uint64_t ub = base0 + (left ? go : 0);
        const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((lim + 15u) & ~(uint64_t)15));
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ub >> 32));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(((uint64_t)bhi << 32) | blo), 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < RW_NP; ++i) SR[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)lane * 16u + (uint32_t)i * 1024u, 0, 0);
        if (++pre_g >= LPL) { pre_g = 0; pre_u += wstride; }
    };
    auto land_group = [&]() {
#pragma unroll
        for (int i = 0; i < RW_NP; ++i) *reinterpret_cast<uint4_t *>(buf + lane * 16 + i * 1024) = SR[i];
    };
    // DIRECT: the counts of this lane's pools (round t: pool t LPL + li of the row's locus) straight from memory into registers, 16 + 8
    // bytes per pool; 16 lanes x 24 bytes = 384 contiguous bytes per row and round.  Two register sets: the next group's loads are in
    // flight while the current group is summed.  Pools past the row's end (the last round) and rounds past the last one point beyond
    // the descriptor's range and read zeros.
    struct RowRegs { uint4_t a[RW_MAXR]; uint2_t b[RW_MAXR]; };
    uint32_t voff[RW_MAXR];
#pragma unroll
    for (int t = 0; t < RW_MAXR; ++t) voff[t] = (t * LPL + li < n) ? (uint32_t)row * rowb + (uint32_t)(t * LPL + li) * 24u : 0x7ffffff0u;
    auto issue_direct = [&](RowRegs &R) {
        const uint64_t go = ((uint64_t)pre_u * 64u + (uint64_t)pre_g * GL) * rowb;
        const uint64_t left = total_bytes > go ? total_bytes - go : 0;
        const uint64_t lim = left < grpb ? left : grpb;
        const uint64_t ub = base0 + (left ? go : 0);
        const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lim);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ub >> 32));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(((uint64_t)bhi << 32) | blo), 0, nrec, 0x00020000);
#pragma unroll
        for (int t = 0; t < RW_MAXR; ++t) {
            R.a[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[t], 0, 0);
            R.b[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[t] + 16u, 0, 0);
        }
        if (++pre_g >= LPL) { pre_g = 0; pre_u += wstride; }
    };
    auto add = [](double a, double b) { return a + b; };
    auto addi = [](int a, int b) { return a + b; };
    auto mini = [](uint32_t a, uint32_t b) { return a < b ? a : b; };

    uint32_t orm_unit = 0;
    int ndirty = 0;
    RowRegs RA;
#if RW_DIRECT_SETS == 2
    RowRegs RB;
#endif
    if constexpr (DIRECT) issue_direct(RA);
    else issue_group();
    for (int64_t unit = wid; unit < nunits; unit += wstride) {
        auto group_body = [&](const RowRegs &R, const int g) { // one of the 64 / GL groups of the unit
            const int lu = g * GL + row;          // this row's locus inside the unit
            const int64_t l = unit * 64 + lu;
            // the six counts of this lane's pool of round t
            auto fetch6 = [&](auto tc, const int pool, uint32_t (&c0)[6]) {
                if constexpr (DIRECT) {
                    constexpr int t = decltype(tc)::value;
                    c0[0] = R.a[t].x; c0[1] = R.a[t].y; c0[2] = R.a[t].z; c0[3] = R.a[t].w; c0[4] = R.b[t].x; c0[5] = R.b[t].y;
                } else {
                    const char *pp = lbase + pool * 24;
                    const uint2_t w0 = *reinterpret_cast<const uint2_t *>(pp), w1 = *reinterpret_cast<const uint2_t *>(pp + 8),
                                  w2 = *reinterpret_cast<const uint2_t *>(pp + 16);
                    c0[0] = w0.x; c0[1] = w0.y; c0[2] = w1.x; c0[3] = w1.y; c0[4] = w2.x; c0[5] = w2.y;
                }
            };
            // ... and the two at byte offsets om, oo of the pool (the survivors of a biallelic locus; pair_at: every row's pair is A, T)
            auto fetch2 = [&](auto tc, const int pool, const int om, const int oo, const bool pair_at, uint32_t &cm, uint32_t &co) {
                if constexpr (DIRECT) {
                    constexpr int t = decltype(tc)::value;
                    if (pair_at) { // (wave-uniform) om, oo are 0 and 4 in some order
                        cm = om ? R.a[t].y : R.a[t].x;
                        co = om ? R.a[t].x : R.a[t].y;
                    } else {
                        const uint32_t c0[6] = {R.a[t].x, R.a[t].y, R.a[t].z, R.a[t].w, R.b[t].x, R.b[t].y};
                        cm = c0[0]; co = c0[0];
#pragma unroll
                        for (int j = 1; j < 6; ++j) { cm = (om == 4 * j) ? c0[j] : cm; co = (oo == 4 * j) ? c0[j] : co; }
                    }
                } else {
                    const char *pp = lbase + pool * 24;
                    cm = *reinterpret_cast<const uint32_t *>(pp + om);
                    co = *reinterpret_cast<const uint32_t *>(pp + oo);
                }
            };
            // every pool of this lane: the full rounds, then the last, partial one
            auto each_pool = [&](auto fn) {
                const int full = n / LPL;
                static_for<0, RW_MAXR>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if (t < full) fn(tc, t * LPL + li);
                    else if (t == full && full * LPL + li < n) fn(tc, full * LPL + li);
                });
            };
            // ---- phase 1: the filter, in SINGLE precision ---------------------------------------------------------------------------------
            // q~_j = sum_i c_ij * (w_i / rs_i) in fp32 is within (n + 4) 2^-24 < 3e-5 of q_j relative (positive terms): it decides every
            // locus whose q stay 2e-4 (relative) away from both thresholds -- practically all -- at a quarter of the fp64 cost; the others
            // get the literal fp64 evaluation below.  q~ == 0 means "no read at all" exactly, whatever the precision.
            float q[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) q[j] = 0.0f;
            uint32_t mincov = 0xffffffffu, orv = 0u;
            int nmiss = 0;
            each_pool([&](auto tc, const int pool) {
                uint32_t c0[6];
                fetch6(tc, pool, c0);
                const float wi = tabf[pool];
                uint32_t rsi = c0[aj(0)], o2 = c0[aj(0)];
#pragma unroll
                for (int j = 1; j < NJ; ++j) { rsi += c0[aj(j)]; o2 |= c0[aj(j)]; }
                orv |= o2;
                mincov = rsi < mincov ? rsi : mincov;
                nmiss += (rsi == 0u) ? 1 : 0;
                const float wr = wi * __builtin_amdgcn_rcpf((float)(rsi > 1u ? rsi : 1u));
#pragma unroll
                for (int j = 0; j < NJ; ++j) q[j] = fmaf((float)c0[aj(j)], wr, q[j]);
            });
            {
                auto addf = [](int x, int y) { return __float_as_int(__int_as_float(x) + __int_as_float(y)); };
#pragma unroll
                for (int j = 0; j < NJ; ++j) q[j] = __int_as_float(row_all<LPL>(__float_as_int(q[j]), addf));
            }
            mincov = row_all<LPL>(mincov, mini);
            {   // missing pools (<= 448) and "a count of 2^29 or more" in one word
                int pk = nmiss | (((orv >> 29) != 0u) ? (1 << 16) : 0);
                pk = row_all<LPL>(pk, addi);
                nmiss = pk & 0xffff;
                orm_unit |= (l < L && (pk >> 16) != 0) ? (1u << 29) : 0u;
            }
            int slotmask = 0;
            bool band = false;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const double qd = (double)q[j];
                slotmask |= !((qd < P.maf) | (qd > (1.00 - P.maf))) ? (1 << j) : 0;
                band = band || (q[j] != 0.0f && (fabs(qd - P.maf) <= 2e-4 * P.maf || fabs(qd - (1.00 - P.maf)) <= 2e-4));
            }
            if (__any(band && l < L)) { // the literal q (sync.rs:258-271) of a locus inside the band: every lane of its row, redundantly
                if (band && l < L) {
                    double qe[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) qe[j] = 0.0;
                    const uint2_t *rp = reinterpret_cast<const uint2_t *>(counts + (size_t)l * (size_t)n * 6); // (its row in memory: rare)
                    for (int i = 0; i < n; ++i) {
                        const uint2_t w0 = rp[3 * i], w1 = rp[3 * i + 1], w2 = rp[3 * i + 2];
                        const uint32_t c0[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
                        uint32_t rs = 0u;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) rs += c0[aj(j)];
                        const double rsd = (double)(rs > 1u ? rs : 1u);
                        const double ri = recip_for_div(rsd);
                        const double wi = tab[i * TW];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) qe[j] = qe[j] + div_by((double)c0[aj(j)], rsd, ri) * wi;
                    }
                    slotmask = 0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        slotmask |= !((qe[j] < P.maf) | (qe[j] > (1.00 - P.maf))) ? (1 << j) : 0;
                        q[j] = (float)qe[j];
                    }
                }
            }
            int nk = 0, keepmask = 0, sa = 0, sb = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool kpj = (slotmask >> j) & 1;
                sa = (kpj && nk == 0) ? j : sa;
                sb = (kpj && nk == 1) ? j : sb;
                nk += kpj ? 1 : 0;
                keepmask |= kpj ? (2 << aj(j)) : 0;
            }
            bool alive = !((double)mincov < P.min_cov);                           // sync.rs:227
            alive = alive && nk >= 2;                                             // sync.rs:284
            alive = alive && nmiss != n;                                          // sync.rs:293
            alive = alive && !(((double)nmiss / (double)n) > P.max_miss);         // sync.rs:297
            alive = alive && l < L;
            bool deferred = alive && nk >= 3; // ols_iter: the joint fit of several alleles: cross products, pool-order sums (second pass)
            int hdr = 0;
            {   // what the lane-per-locus streaming pass could not close from clean sums: told to the host, which picks the kernel of the
                // NEXT batch by it (run_locus_op)
                bool stray = nk >= 3;
#pragma unroll
                for (int j = 0; j < NJ; ++j) stray = stray || (!((slotmask >> j) & 1) && q[j] != 0.0f);
                ndirty += (alive && stray && li == 0) ? 1 : 0;
            }
            if constexpr (OP == OP_CHISQ) {
                // ---- chisq_test: the table of the SURVIVORS' frequencies, any number of them (tables/chisq_test.rs:5-47) -------------------
                // chi2 = total * (sum_j A_j / cs_j - 1), A_j = sum_i f_ij^2 / rowsum_i, total = sum_i rowsum_i; a covered pool's row sum is
                // 1 to an ulp, so A_j = sum f^2 and total = the covered pools (statistics are compared at 1e-10).  Nothing is deferred.
                deferred = false;
                if (__any(alive)) {
                    double csj[NJ], ddj[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { csj[j] = 0.0; ddj[j] = 0.0; }
                    int nmiss2 = 0;
                    const bool two = nk == 2;                      // this row's locus (the usual case): only the pair's two columns
                    const bool anygen = __any(alive && nk >= 3);   // (a locus' arithmetic depends on that locus alone: bits do not change with its neighbours)
                    // two survivors: only the RARER one (by phase 1's q~) is summed; f_a + f_b = 1 in every covered pool gives the other's
                    // sums without cancellation (they are of the order of the pool count)
                    int oa = 0, ob = 0;
                    float qa = q[0], qb = q[0];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        oa = (sa == j) ? 4 * aj(j) : oa; ob = (sb == j) ? 4 * aj(j) : ob;
                        qa = (sa == j) ? q[j] : qa; qb = (sb == j) ? q[j] : qb;
                    }
                    const int om = (qb < qa) ? ob : oa, oo = (qb < qa) ? oa : ob;
                    const bool pair_at = __all(!(alive && two) || (oa + ob == 4 && (oa == 0 || ob == 0))); // (DIRECT: the pair is A, T in every row)
                    each_pool([&](auto tc, const int pool) {
                        if (two) {
                            uint32_t cm, co;
                            fetch2(tc, pool, om, oo, pair_at, cm, co);
                            const uint32_t rs2 = cm + co;
                            nmiss2 += (rs2 == 0u) ? 1 : 0;
                            const double rsd2 = (double)(rs2 > 1u ? rs2 : 1u);
                            const double r0 = __builtin_amdgcn_rcp(rsd2);
                            const double f = (double)cm * fma(fma(-rsd2, r0, 1.0), r0, r0);
                            csj[0] += f;
                            ddj[0] = fma(f, f, ddj[0]);
                        } else {
                            uint32_t c0[6];
                            fetch6(tc, pool, c0);
                            uint32_t cj[NJ], rs2 = 0u;
#pragma unroll
                            for (int j = 0; j < NJ; ++j) { cj[j] = ((slotmask >> j) & 1) ? c0[aj(j)] : 0u; rs2 += cj[j]; }
                            nmiss2 += (rs2 == 0u) ? 1 : 0;
                            const double rsd2 = (double)(rs2 > 1u ? rs2 : 1u);
                            const double r0 = __builtin_amdgcn_rcp(rsd2);
                            const double r1 = fma(fma(-rsd2, r0, 1.0), r0, r0);
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                const double f = (double)cj[j] * r1;
                                csj[j] += f;
                                ddj[j] = fma(f, f, ddj[j]);
                            }
                        }
                    });
                    nmiss2 = row_all<LPL>(nmiss2, addi);
                    double acc = 0.0;
                    csj[0] = row_all<LPL>(csj[0], add);
                    ddj[0] = row_all<LPL>(ddj[0], add);
                    if (anygen) {
#pragma unroll
                        for (int j = 1; j < NJ; ++j) { csj[j] = row_all<LPL>(csj[j], add); ddj[j] = row_all<LPL>(ddj[j], add); }
                    }
                    if (two) {
                        const double ncov = (double)(n - nmiss2);
                        acc = ddj[0] / csj[0] + (ncov - 2.0 * csj[0] + ddj[0]) / (ncov - csj[0]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc = ((slotmask >> j) & 1) ? acc + ddj[j] / csj[j] : acc;
                    }
                    if (alive) {
                        int idsp = 0, r = 0;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const bool kpj = (slotmask >> j) & 1;
                            idsp |= kpj ? (aj(j) << (3 * r)) : 0;
                            r += kpj ? 1 : 0;
                        }
                        hdr = 1 | (nmiss2 > 0 ? 2 : 0) | (nk << 4) | (idsp << 8);
                        if (li == 0) { sums[lu * NSUM] = acc; sums[lu * NSUM + 1] = (double)(n - nmiss2); }
                    }
                }
            }
            // ---- phase 2: exactly two survivors a < b -- frequencies over THEIR coverage (gwas/ols.rs:210-230 -> sync.rs:166-192) ---------
            // The design column is the MINOR allele (stable sort by decreasing column sum, sync.rs:477-506; the major one is dropped,
            // ols.rs:227-230).  The sums are taken for the allele phase 1 saw as the rarer one; f_a + f_b = 1 in every covered pool, so
            // the other allele's sums follow from them should the column sums say otherwise (both are then near 1/2: no cancellation).
            if (OP == OP_OLS && __any(alive && nk == 2)) {
                float qa = q[0], qb = q[0];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { qa = (sa == j) ? q[j] : qa; qb = (sb == j) ? q[j] : qb; }
                const bool m_is_b = qb < qa;                     // the allele summed: m
                int oa = 0, ob = 0;                              // byte offsets of the survivors' counts inside a pool
#pragma unroll
                for (int j = 0; j < NJ; ++j) { oa = (sa == j) ? 4 * aj(j) : oa; ob = (sb == j) ? 4 * aj(j) : ob; }
                const int om = m_is_b ? ob : oa, oo = m_is_b ? oa : ob;
                double csm = 0.0, ddm = 0.0, xym[K];
#pragma unroll
                for (int t = 0; t < K; ++t) xym[t] = 0.0;
                int nmiss2 = 0;
                const bool pair_at = __all(!(alive && nk == 2) || (oa + ob == 4 && (oa == 0 || ob == 0))); // (DIRECT: the pair is A, T in every row)
                each_pool([&](auto tc, const int pool) {
                    uint32_t cm, co;
                    fetch2(tc, pool, om, oo, pair_at, cm, co);
                    const uint32_t rs2 = cm + co;
                    nmiss2 += (rs2 == 0u) ? 1 : 0;
                    // c / rs to a few ulp: the hardware reciprocal and ONE Newton step (these sums need 1e-10, not the last bit)
                    const double rsd2 = (double)(rs2 > 1u ? rs2 : 1u);
                    const double r0 = __builtin_amdgcn_rcp(rsd2);
                    const double f = (double)cm * fma(fma(-rsd2, r0, 1.0), r0, r0);
                    csm += f;
                    ddm = fma(f, f, ddm);
#pragma unroll
                    for (int tt = 0; tt < K; ++tt) xym[tt] = fma(f, tab[pool * TW + 1 + tt], xym[tt]);
                });
                csm = row_all<LPL>(csm, add);
                ddm = row_all<LPL>(ddm, add);
#pragma unroll
                for (int t = 0; t < K; ++t) xym[t] = row_all<LPL>(xym[t], add);
                nmiss2 = row_all<LPL>(nmiss2, addi);
                if (alive && nk == 2) {
                    const double ncov = (double)(n - nmiss2);
                    const double cso = ncov - csm;                              // the other survivor's column sum
                    // decisions that could depend on the ORDER of the sums go to the pool-order second pass: which allele is the major
                    // one (column sums within 1e-9 n), a design within 1e-8 of singular (ols.rs:77-83), derived sums with uncovered pools
                    const bool close_call = fabs(csm - cso) <= 1e-9 * (double)n;
                    const bool swap = cso < csm;                                 // phase 1's guess was the major allele after all
                    double csd = csm, ddd = ddm, xyd[K];
#pragma unroll
                    for (int t = 0; t < K; ++t) xyd[t] = xym[t];
                    if (swap) {
                        csd = cso;
                        ddd = ncov - 2.0 * csm + ddm;
#pragma unroll
                        for (int t = 0; t < K; ++t) xyd[t] = P.sy[t] - xym[t];
                    }
                    const double det = (double)n * ddd - csd * csd;                 // of the 2 x 2 normal matrix, up to rounding
                    const bool near_singular = !(det > 1e-8 * (double)n * ddd);
                    deferred = close_call || near_singular || (swap && nmiss2 > 0);
                    const bool d_is_b = swap ? !m_is_b : m_is_b;
                    int idc = 0;
                    const int dslot = d_is_b ? sb : sa;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) idc = (dslot == j) ? aj(j) : idc;
                    hdr = 1 | (nmiss2 > 0 ? 2 : 0) | (idc << 4);
                    if (li == 0 && !deferred) {
                        sums[lu * NSUM] = csd;
                        sums[lu * NSUM + 1] = ddd;
#pragma unroll
                        for (int tt = 0; tt < K; ++tt) sums[lu * NSUM + 2 + tt] = xyd[tt];
                    }
                }
            }
            if (li == 0) {
                hdrs[lu] = deferred ? (1 << 30) : hdr; // bit 0: close in place | 1: a pool uncovered over the survivors | 4..6 allele (chisq_test: survivors, 8.. their ids) | 30: listed
                if (deferred) rec_flags[l] = FLAG_ALIVE | keepmask | FLAG_SECOND | (nk << H_NK_SHIFT);
            }
        };
        if constexpr (DIRECT) {
#if RW_DIRECT_SETS == 2
            for (int g = 0; g < LPL; g += 2) {    // (RA holds group g; the set just summed takes the loads of the group after next)
                issue_direct(RB);
                group_body(RA, g);
                issue_direct(RA);
                group_body(RB, g + 1);
            }
#else
            for (int g = 0; g < LPL; ++g) {       // (one set: the other waves of the SIMD cover the wait)
                group_body(RA, g);
                issue_direct(RA);
            }
#endif
        } else {
            for (int g = 0; g < LPL; ++g) {
                __builtin_amdgcn_wave_barrier();
                land_group();
                __builtin_amdgcn_wave_barrier();
                issue_group();
                group_body(RA, g);
            }
        }
        // ---- the unit's 64 loci: lane = locus -- close, list, write -------------------------------------------------------------------
        __builtin_amdgcn_wave_barrier();
        {
            const int64_t l = unit * 64 + lane;
            const int hdr = hdrs[lane];
            const bool listed = (hdr & (1 << 30)) != 0;
            const unsigned long long bal = __ballot(listed);
            if (bal) {
                unsigned long long basev = 0;
                if (lane == 0) basev = atomicAdd(second_count + SC_LIST, (unsigned long long)__popcll(bal));
                basev = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(basev >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)basev);
                if (listed) second[basev + __popcll(bal & ((1ull << lane) - 1ull))] = l;
            }
            const bool simple = (hdr & 1) != 0;
            int nout = 0, idsp = 0;
            double mf = NAN, st[K], pv[K];
#pragma unroll
            for (int t = 0; t < K; ++t) { st[t] = NAN; pv[t] = NAN; }
            if constexpr (OP == OP_CHISQ) {
                if (__any(simple)) {
                    const double pz = (hdr & 2) ? NAN : 0.0;
                    const int nk = (hdr >> 4) & 7;
                    const double chi2 = sums[lane * NSUM + 1] * (sums[lane * NSUM] - 1.0) + pz; // tables/chisq_test.rs:15-31 regrouped
                    const double df = (double)(n * nk) - 1.0;
                    const double p = pg_chisq_upper_p(chi2, df, pg_ln_gamma(df / 2.0)); // :33-35
                    if (simple) { st[0] = chi2; pv[0] = p; nout = nk; idsp = (hdr >> 8) & 0x7fff; }
                }
            } else if (__any(simple)) {
                const double pz = (hdr & 2) ? NAN : 0.0; // a pool uncovered over the survivors: NaN frequencies, NaN sums (sync.rs:176-183)
                double cs1[1], xx1[1][1], xy1[1][MAXK], b1[MAXK][1], p1[MAXK][1];
                cs1[0] = sums[lane * NSUM] + pz;
                xx1[0][0] = sums[lane * NSUM + 1] + pz;
#pragma unroll
                for (int t = 0; t < MAXK; ++t) xy1[0][t] = (t < K) ? sums[lane * NSUM + 2 + (t < K ? t : 0)] + pz : 0.0;
                bool singular;
                ols_solve<2>(cs1, xx1, xy1, K, P, tcoef, singular, b1, p1);
                if (simple && !singular) { // Err -> the whole locus is dropped (ols.rs:250-253)
                    nout = 1;
                    idsp = (hdr >> 4) & 7;
                    mf = cs1[0] / (double)n; // ols.rs:266
#pragma unroll
                    for (int t = 0; t < K; ++t) { st[t] = b1[t][0]; pv[t] = p1[t][0]; }
                }
            }
            // (a listed locus gets the "dropped" pattern here; the second pass overwrites it later in the stream)
            char *r = stage + (size_t)lane * RECB;
            *reinterpret_cast<uint2_t *>(r) = uint2_t{(uint32_t)nout, (uint32_t)idsp};
            *reinterpret_cast<double *>(r + 8) = mf;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                *reinterpret_cast<double *>(r + 16 + 16 * t) = st[t];
                *reinterpret_cast<double *>(r + 24 + 16 * t) = pv[t];
            }
        }
        __builtin_amdgcn_wave_barrier();
        {   // slot 0 of every output array, 64 consecutive loci: contiguous runs
            const int64_t l0 = unit * 64;
            const int nv = (int)((L - l0) < 64 ? (L - l0) : 64);
            auto hdr_of = [&](int j) { return *reinterpret_cast<const uint2_t *>(stage + (size_t)j * RECB); };
            auto dbl_of = [&](int j, int off) { return *reinterpret_cast<const double *>(stage + (size_t)j * RECB + off); };
            if constexpr (OP == OP_CHISQ) {
                if (lane < nv) { // n_out, the surviving alleles in the slots below it (column order), chi2, p: plain [L] arrays
                    const uint2_t h = hdr_of(lane);
                    const int no = (int)h.x;
                    O.n_out[l0 + lane] = no;
#pragma unroll
                    for (int r = 0; r < PG_MAX_OUT; ++r)
                        if (r < no) O.ids[(size_t)r * (size_t)P.L + (size_t)(l0 + lane)] = (int)((h.y >> (3 * r)) & 7u);
                    O.stat[l0 + lane] = dbl_of(lane, 16);
                    O.pv[l0 + lane] = dbl_of(lane, 24);
                }
            } else if (lane < nv) {
                const uint2_t h = hdr_of(lane);
                if (coalesced) {
                    __builtin_nontemporal_store((int32_t)h.x, O.n_out + l0 + lane);
                    __builtin_nontemporal_store(h.x ? (int32_t)(h.y & 7u) : -1, O.ids + l0 + lane);
                    __builtin_nontemporal_store(h.x ? dbl_of(lane, 8) : NAN, O.mf + l0 + lane);
                } else if (P.t0 == 0) {
                    O.n_out[l0 + lane] = (int32_t)h.x;
                    O.ids[l0 + lane] = h.x ? (int32_t)(h.y & 7u) : -1;
                    O.mf[l0 + lane] = h.x ? dbl_of(lane, 8) : NAN;
                }
            }
            // stat / pval: [slot 0][locus][trait of the call]
            for (int e = lane; OP == OP_OLS && e < nv * K; e += 64) {
                const int j = e / K, t = e - j * K;
                const bool on = hdr_of(j).x != 0;
                const size_t o = (size_t)(l0 + j) * P.k_total + P.t0 + t;
                O.stat[o] = on ? dbl_of(j, 16 + 16 * t) : NAN;
                O.pv[o] = on ? dbl_of(j, 24 + 16 * t) : NAN;
            }
        }
        if (__any((orm_unit >> 29) != 0u)) { // a count the 32-bit coverage sums cannot take: the host reports it
            if (lane == 0) atomicOr(second_count + SC_COMPLAINT, 1ull);
            orm_unit = 0;
        }
    }
    {
        int c = ndirty;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0 && c) atomicAdd(second_count + SC_DIRTY, (unsigned long long)c);
    }
}

// ---- second pass: only the loci the first pass listed ----------------------------------------------
// ---- second pass: only the loci the first pass listed ----------------------------------------------
// One lane per listed locus (rows gathered through the list).  The survivors are known now (flags), so the sums are taken over
// the frequencies of the FILTERED counts, as the reference does (gwas/ols.rs:210-230 -> sync.rs:166-192).  Round 4:
//  * the surviving columns are COMPACTED (NS running columns instead of all six: 3 / 6 / 10 / 15 cross products for 2 / 3 / 4 / 5
//    survivors instead of 21) and the code is specialised on the largest survivor count of the wave;
//  * rows still arrive through a wave-private LDS tile in stages of eight pools (48 lanes fetch 4 rows x 192 contiguous bytes per
//    instruction: a lane reading its own row 8 bytes at a time makes every load instruction touch 64 cache lines and the pass ran
//    at 1 TB/s, measured), but the NEXT stage's loads are in flight while a stage is summed, and the eight pools of a stage are
//    straight-line code (no branch inside: with one, the wait-count pass waits for every load at every pool).
template <int OP, int NS, int K, int PB>
__device__ __forceinline__ void second_tile(const uint32_t *__restrict__ counts, const double *__restrict__ Y, int32_t *rec_flags,
                                            double *rec, char *tile, const int64_t *lidx, const int64_t l, const int mask,
                                            const bool valid, const LocusParams &P, const int lane) {
    const int n = P.n;
    // the r-th surviving sync column (ascending); columns r >= nk read as zero counts and stay out of every result
    int col[NS];
    bool kp[NS];
    {
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < NS; ++r) col[r] = 7;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const bool k = (mask & (2 << j)) != 0;
#pragma unroll
            for (int r = 0; r < NS; ++r) col[r] = (k && cnt == r) ? j : col[r];
            cnt += k ? 1 : 0;
        }
#pragma unroll
        for (int r = 0; r < NS; ++r) kp[r] = r < cnt;
    }
    Sums<OP, NS, K> A;
    A.clear();
    int n_missing = 0;
    const char *row = tile + lane * LO_PITCH;
    const int nfull = n / LO_CHP;
    const int nst = (n + LO_CHP - 1) / LO_CHP;
    auto rowsel = [&](int r) { return lidx[r]; };
    // (one stage of loads in flight while a stage is summed; two were measured: no gain on short lists, -4 % on long ones)
    StageRegs S;
    auto load_stage = [&](int st) {
        const int pool0 = st * LO_CHP;
        if (st < nfull) stage_load<PB, false>(S, counts, n, pool0, LO_CHP, lane, rowsel);
        else stage_load<8, true>(S, counts, n, pool0, n - pool0, lane, rowsel);
    };
    load_stage(0);
    for (int st = 0; st < nst; ++st) {
        const int pool0 = st * LO_CHP;
        const int np = st < nfull ? LO_CHP : n - pool0;
        if (st < nfull) stage_store<PB, false>(S, tile, np, lane);
        else stage_store<8, true>(S, tile, np, lane);
        __builtin_amdgcn_wave_barrier();
        if (st + 1 < nst) load_stage(st + 1); // in flight while this stage is summed
        // The eight pools of a stage are a ROLLED loop (two per trip, the next trip's words requested before this trip's arithmetic):
        // unrolled eight times, the bodies of the NS variants together with the closing code overflowed the instruction cache that
        // two CUs share, and every wave waited for instruction fetches (19 us per stage of eight pools, measured, against 3 us of
        // arithmetic).  Pools past the row's end in the last stage are stale tile bytes, forced to zero counts.
        uint2_t wn[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) wn[u][k] = *reinterpret_cast<const uint2_t *>(row + u * 24 + 8 * k);
#pragma unroll 1
        for (int a0 = 0; a0 < LO_CHP; a0 += 2) {
            uint2_t w[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k) w[u][k] = wn[u][k];
            {
                const int an = (a0 + 2 < LO_CHP) ? a0 + 2 : a0; // (the last trip re-reads its own words)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k = 0; k < 3; ++k) wn[u][k] = *reinterpret_cast<const uint2_t *>(row + (an + u) * 24 + 8 * k);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int a = a0 + u;
                const bool live = a < np;
                const uint32_t c[NA] = {w[u][0].x, w[u][0].y, w[u][1].x, w[u][1].y, w[u][2].x, w[u][2].y};
                uint32_t cc[NS];
                uint32_t rsi = 0u; // row sum over the surviving alleles (second to_frequencies, sync.rs:170-175): exact in integers
#pragma unroll
                for (int r = 0; r < NS; ++r) {
                    uint32_t v = 0u;
#pragma unroll
                    for (int j = 0; j < NA; ++j) v = (col[r] == j) ? c[j] : v;
                    v = live ? v : 0u;
                    cc[r] = v;
                    rsi += v;
                }
                const bool rowok = rsi != 0u;
                const double rsd = (double)(rsi > 1u ? rsi : 1u);
                const double rinv = recip_for_div(rsd);
                n_missing += (live && !rowok) ? 1 : 0;
                double f[NS];
#pragma unroll
                for (int r = 0; r < NS; ++r) f[r] = div_by((double)cc[r], rsd, rinv);
                A.add_pool(f, rowok, Y + (size_t)(pool0 + a) * K); // (the host pads Y with LO_CHP rows of zeros)
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    auto ajc = [&](int r) { return col[r] & 7; };
    emit_record<OP, NS, K>(A, n_missing > 0, kp, true, false, valid, rec_flags, rec, unit_slot(l, P.pshift), ajc, P.sort_desc != 0);
}

// The list grouped by the number of surviving alleles: group b (b + 2 survivors) occupies whole tiles of 64 entries of `sorted`, so
// that a wave of the sums / closing kernels below runs code compiled for ONE number of survivors (mixed tiles ran every lane at the
// tile's maximum and, in the closing, every variant in turn: together with the instruction-cache misses of that much code, 2.5 x).
__device__ __forceinline__ void group_tiles(const unsigned long long *__restrict__ second_count, int64_t (&cnt)[LO_NB], int64_t (&first)[LO_NB + 1]) {
    first[0] = 0;
#pragma unroll
    for (int b = 0; b < LO_NB; ++b) {
        cnt[b] = (int64_t)second_count[b];
        first[b + 1] = first[b] + (cnt[b] + 63) / 64;
    }
}

// how many listed loci keep 2, 3, .. alleles (five atomics per wave)
__global__ __launch_bounds__(256) void k_locus_hist(const int64_t *__restrict__ second, unsigned long long *second_count,
                                                    const int32_t *__restrict__ rec_flags, int64_t total, int pshift) {
    const int lane = threadIdx.x & 63;
    int c[LO_NB];
#pragma unroll
    for (int b = 0; b < LO_NB; ++b) c[b] = 0;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t * 64 < total; t += nw) {
        const int64_t e = t * 64 + lane;
        const int nk = e < total ? ((rec_flags[unit_slot(second[e], pshift)] >> H_NK_SHIFT) & 7) : 0;
#pragma unroll
        for (int b = 0; b < LO_NB; ++b) c[b] += __popcll(__ballot(nk == b + 2));
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < LO_NB; ++b)
            if (c[b]) atomicAdd(second_count + b, (unsigned long long)c[b]);
    }
}

constexpr int SORT_CH = 16; // tiles per wave and reservation
__global__ __launch_bounds__(256) void k_locus_sort(const int64_t *__restrict__ second, unsigned long long *second_count,
                                                    const int32_t *__restrict__ rec_flags, int64_t *__restrict__ sorted, int64_t total,
                                                    int pshift) {
    const int lane = threadIdx.x & 63;
    int64_t cnt[LO_NB], first[LO_NB + 1];
    group_tiles(second_count, cnt, first);
    const int64_t ntile = (total + 63) / 64;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SORT_CH; t0 < ntile; t0 += nw * SORT_CH) {
        // this wave's chunk: count per group, reserve once per group, scatter
        int c[LO_NB];
        int nkv[SORT_CH];
        int64_t lv[SORT_CH];
#pragma unroll
        for (int b = 0; b < LO_NB; ++b) c[b] = 0;
#pragma unroll
        for (int u = 0; u < SORT_CH; ++u) {
            const int64_t e = (t0 + u) * 64 + lane;
            const bool valid = e < total;
            lv[u] = valid ? second[e] : 0;
            nkv[u] = valid ? ((rec_flags[unit_slot(lv[u], pshift)] >> H_NK_SHIFT) & 7) : 0;
#pragma unroll
            for (int b = 0; b < LO_NB; ++b) c[b] += __popcll(__ballot(nkv[u] == b + 2));
        }
        int64_t base[LO_NB];
#pragma unroll
        for (int b = 0; b < LO_NB; ++b) {
            unsigned long long bv = 0;
            if (lane == 0 && c[b]) bv = atomicAdd(second_count + SC_CURSOR + b, (unsigned long long)c[b]);
            bv = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bv >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
            base[b] = first[b] * 64 + (int64_t)bv;
        }
#pragma unroll
        for (int u = 0; u < SORT_CH; ++u) {
#pragma unroll
            for (int b = 0; b < LO_NB; ++b) {
                const bool mine = nkv[u] == b + 2;
                const unsigned long long bal = __ballot(mine);
                if (mine) sorted[base[b] + __popcll(bal & ((1ull << lane) - 1ull))] = lv[u];
                base[b] += __popcll(bal);
            }
        }
    }
}

// the tile t of the list: this lane's entry (clamped to the last one: `valid` false) and the number of survivors the tile's code is
// compiled for -- the group's when the list is grouped (k_locus_sort), the largest of the tile's loci when it is not (short lists)
__device__ __forceinline__ int tile_entry(const int64_t t, const int grouped, const int64_t total, const int64_t (&cnt)[LO_NB],
                                          const int64_t (&first)[LO_NB + 1], const int64_t *__restrict__ list,
                                          const int32_t *__restrict__ rec_flags, const int pshift, const int lane, int64_t &l, bool &valid) {
    if (grouped) {
        int b = 0;
#pragma unroll
        for (int q = 1; q < LO_NB; ++q) b = (t >= first[q]) ? q : b;
        b = __builtin_amdgcn_readfirstlane(b);
        int64_t cb = cnt[0], fb = first[0];
#pragma unroll
        for (int q = 1; q < LO_NB; ++q) { cb = (b == q) ? cnt[q] : cb; fb = (b == q) ? first[q] : fb; }
        const int64_t e = (t - fb) * 64 + lane;
        valid = e < cb;
        l = list[fb * 64 + (valid ? e : cb - 1)];
        return b + 2;
    }
    const int64_t e = t * 64 + lane;
    valid = e < total;
    l = list[valid ? e : total - 1];
    int nsw = (rec_flags[unit_slot(l, pshift)] >> H_NK_SHIFT) & 7;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nsw = max(nsw, __shfl_xor(nsw, off));
    return __builtin_amdgcn_readfirstlane(nsw);
}

template <int OP, int NMAX, int K, int PB>
__global__ __launch_bounds__(LO_THREADS) void k_locus_second(
    const uint32_t *__restrict__ counts, const double *__restrict__ Y, int32_t *rec_flags,
    double *rec, const int64_t *__restrict__ list,
    const unsigned long long *__restrict__ second_count, const LocusParams P, const int grouped, const int64_t total) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    char *tile = lds_raw + wave * LO_TILEB;
    int64_t *lidx = reinterpret_cast<int64_t *>(lds_raw + LO_WAVES * LO_TILEB) + wave * 64;
    int64_t cnt[LO_NB], first[LO_NB + 1];
    group_tiles(second_count, cnt, first);
    const int64_t ntile = grouped ? first[LO_NB] : (total + 63) / 64;
    const int64_t wstride = (int64_t)gridDim.x * LO_WAVES;
    for (int64_t t = (int64_t)blockIdx.x * LO_WAVES + wave; t < ntile; t += wstride) {
        int64_t l;
        bool valid;
        const int ns = tile_entry(t, grouped, total, cnt, first, list, rec_flags, P.pshift, lane, l, valid);
        __builtin_amdgcn_wave_barrier();
        lidx[lane] = l;
        __builtin_amdgcn_wave_barrier();
        const int mask = rec_flags[unit_slot(l, P.pshift)];
        // (the arithmetic of a locus does not depend on the variant: padding columns contribute exact zeros to sums that are never read)
        if (ns <= 2) second_tile<OP, 2, K, PB>(counts, Y, rec_flags, rec, tile, lidx, l, mask, valid, P, lane);
        else if (ns == 3) second_tile<OP, 3, K, PB>(counts, Y, rec_flags, rec, tile, lidx, l, mask, valid, P, lane);
        else if (ns == 4) second_tile<OP, 4, K, PB>(counts, Y, rec_flags, rec, tile, lidx, l, mask, valid, P, lane);
        else if (ns == 5 || NMAX == 5) second_tile<OP, 5, K, PB>(counts, Y, rec_flags, rec, tile, lidx, l, mask, valid, P, lane);
        else second_tile<OP, NMAX, K, PB>(counts, Y, rec_flags, rec, tile, lidx, l, mask, valid, P, lane);
    }
}

// closing arithmetic of the listed loci from their records: its own launch (the sums kernel's code stays small); on a grouped list a
// tile = one number of survivors = one variant of the closing code per wave
template <int OP>
__global__ __launch_bounds__(256) void k_locus_close(const int32_t *rec_flags, const double *rec, const int64_t *__restrict__ list,
                                                     const unsigned long long *__restrict__ second_count,
                                                     const double *__restrict__ tcoef, const StreamOut O, const LocusParams P,
                                                     const int grouped, const int64_t total) {
    const int lane = threadIdx.x & 63;
    int64_t cnt[LO_NB], first[LO_NB + 1];
    group_tiles(second_count, cnt, first);
    const int64_t ntile = grouped ? first[LO_NB] : (total + 63) / 64;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntile; t += nw) {
        int64_t l;
        bool valid;
        const int ns = tile_entry(t, grouped, total, cnt, first, list, rec_flags, P.pshift, lane, l, valid);
        if (!valid) continue;
        if (!grouped) close_locus<OP, 0>(l, rec_flags, rec, tcoef, O.n_out, O.ids, O.mf, O.stat, O.pv, P);
        else if (ns == 2) close_locus<OP, 2>(l, rec_flags, rec, tcoef, O.n_out, O.ids, O.mf, O.stat, O.pv, P);
        else if (ns == 3) close_locus<OP, 3>(l, rec_flags, rec, tcoef, O.n_out, O.ids, O.mf, O.stat, O.pv, P);
        else if (ns == 4) close_locus<OP, 4>(l, rec_flags, rec, tcoef, O.n_out, O.ids, O.mf, O.stat, O.pv, P);
        else if (ns == 5) close_locus<OP, 5>(l, rec_flags, rec, tcoef, O.n_out, O.ids, O.mf, O.stat, O.pv, P);
        else close_locus<OP, 6>(l, rec_flags, rec, tcoef, O.n_out, O.ids, O.mf, O.stat, O.pv, P);
    }
}


// ---------------------------------------------------------------------------------------------
// The passes of one launch group (a trait pair, or the loader's plan): the streaming pass over every locus, then the second
// pass + closing of the loci it listed (both leave at once when the list is empty).
inline int stream_period(int n) { // loci per lane: the smallest count whose bytes are whole 128-byte lines
    int period = 1;
    while ((((int64_t)n * 24 * period) & 127) != 0) period *= 2;
    return period;
}

struct StreamWs { // device pointers into the context's workspace
    double *table;              // n x TW: w_i, y_i0, ... for the streaming pass
    double *Y;                  // n x MAXK for the second pass
    double *tcoef;
    double *rec;                // records of the second pass, one per locus (touched only for listed loci)
    int64_t *second;            // the second pass' list (L entries), and grouped by the number of survivors (L + 64 LO_NB)
    int64_t *sorted;
    unsigned long long *second_count; // SC_WORDS words
    int32_t *flags;
};

template <int OP>
int launch_passes(pg_ctx *ctx, int kid, const uint32_t *counts_dev, const StreamWs &W, const StreamOut &O, const LocusParams &P, int kg, bool rns,
                  int64_t *listed, bool *complaint) {
    const int n = P.n;
    const int64_t L = P.L;
    const int M = stream_period(n);
    const int TW = (OP == OP_OLS || OP == OP_PEARSON) ? 1 + kg : 1;
    const int recb = (OP == OP_LOAD) ? 4 : 16 + 16 * kg;
    int staged = (size_t)64 * M * recb <= (size_t)ST_STAGE ? 1 : 0; // bit 0: results staged per unit in LDS; bit 1: written as 16-byte pieces
    if (staged && OP != OP_LOAD) {
        auto al16 = [](const void *q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        if (P.k_total == kg && P.t0 == 0 && al16(O.n_out) && al16(O.ids) && al16(O.mf) && al16(O.stat) && al16(O.pv)) staged |= 2;
    }
    const size_t shmem = (size_t)LO_WAVES * (ST_SLOTB + ST_STAGE) + sizeof(double) * (size_t)TW * n;
    PG_CHECK(ctx, shmem <= 160 * 1024, "locus op: too many pools (%d) for the pool table in LDS", n);
    const int64_t nunits = (L + (int64_t)64 * M - 1) / ((int64_t)64 * M);
    const int64_t blocks = (nunits + LO_WAVES - 1) / LO_WAVES;
    const int64_t cap = (int64_t)ctx->cus * LO_BLOCKS_DEF; // the resident blocks: one long sequence of units per wave
    const int grid = (int)(blocks < cap ? blocks : cap);
    auto pick = [&]() -> const void * {
        if (OP == OP_OLS && kg == 2)
            return rns ? (const void *)k_locus_stream<OP, true, (OP == OP_OLS ? 2 : 1)> : (const void *)k_locus_stream<OP, false, (OP == OP_OLS ? 2 : 1)>;
        return rns ? (const void *)k_locus_stream<OP, true, 1> : (const void *)k_locus_stream<OP, false, 1>;
    };
    const void *kstream = pick();
    PG_HIP(ctx, hipFuncSetAttribute(kstream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    PG_HIP(ctx, hipMemsetAsync(W.second_count, 0, 8 * SC_WORDS, ctx->stream)); // the list's length and groups, the sort's cursors, the streaming pass' complaint flag
    // ols_iter from 32 pools up: the order-free kernel (a locus per row of 16 / 32 / 64 lanes, coalesced reads) or the streaming pass
    // (sums in pool order, mean frequencies bit-identical to the reference's)
    int lpl = 0;
    // Which kernel: the order-free one costs the same whatever the counts look like (0.63 of the HBM peak at 100 pools); the
    // streaming pass is faster on clean counts (0.68 - 0.70) and slower on error-bearing ones (second pass: 0.53).  A context
    // remembers what its last ols_iter batch looked like (pieces of one file look alike) and starts with the robust kernel.
    // POOLGEN_OLS_ITER_KERNEL=rows|stream fixes the choice (tests that compare bits across calls; A/B runs).
    constexpr int ROWS_OP = (OP == OP_CHISQ) ? 1 : 0;
    bool &rows_next = ctx->rows_next[ROWS_OP];
    if (P.t0 == 0) ctx->rows_call[ROWS_OP] = rows_next; // one choice per call: its launch groups (trait pairs) must agree on who wrote slot 0
    bool want_rows = ctx->rows_call[ROWS_OP];
    if (const char *e = std::getenv("POOLGEN_OLS_ITER_KERNEL")) want_rows = std::strcmp(e, "stream") != 0;
    if ((OP == OP_OLS || OP == OP_CHISQ) && n >= 32 && want_rows) {
        lpl = n <= 112 ? 16 : (n <= 224 ? 32 : (n <= 448 ? 64 : 0));
        if (lpl == 64 && (n & 1)) lpl = 0; // a group = one locus must be a whole number of 16-byte pieces
    }
    if (kid >= 0) pg_prof_begin(ctx, kid);
    if (lpl) {
        if constexpr (OP == OP_OLS || OP == OP_CHISQ) {
            // chisq_test reads its pools straight into registers (DIRECT: no staging buffer, three waves per SIMD): -3 % at 100 pools,
            // -10 % at 200 on the same box; ols_iter gains nothing from it (measured: DESIGN section 3.3) and keeps the buffer.
            // POOLGEN_ROWS_DIRECT=0 puts chisq_test back on the buffer (A/B runs).
            bool direct = OP == OP_CHISQ;
            if (const char *e = std::getenv("POOLGEN_ROWS_DIRECT")) direct = direct && std::strcmp(e, "0") != 0;
            auto pick_rows = [&]() -> const void * {
                if constexpr (OP == OP_CHISQ) {
#define PG_ROWS(LPLV) (direct ? (rns ? (const void *)k_ols_rows<OP, LPLV, true, 1, true> : (const void *)k_ols_rows<OP, LPLV, false, 1, true>) \
                              : (rns ? (const void *)k_ols_rows<OP, LPLV, true, 1, false> : (const void *)k_ols_rows<OP, LPLV, false, 1, false>))
                    return lpl == 16 ? PG_ROWS(16) : (lpl == 32 ? PG_ROWS(32) : PG_ROWS(64));
#undef PG_ROWS
                } else {
#define PG_ROWS(LPLV) (kg == 2 ? (rns ? (const void *)k_ols_rows<OP, LPLV, true, 2, false> : (const void *)k_ols_rows<OP, LPLV, false, 2, false>) \
                               : (rns ? (const void *)k_ols_rows<OP, LPLV, true, 1, false> : (const void *)k_ols_rows<OP, LPLV, false, 1, false>))
                    return lpl == 16 ? PG_ROWS(16) : (lpl == 32 ? PG_ROWS(32) : PG_ROWS(64));
#undef PG_ROWS
                }
            };
            const void *krows = pick_rows();
            const size_t per_wave = (direct ? 0 : (size_t)RW_BUF) + 64 * (2 + kg) * 8 + 64 * 4 + 64 * (16 + 16 * kg);
            const size_t shr = (size_t)LO_WAVES * per_wave + sizeof(double) * (size_t)TW * n + sizeof(float) * (size_t)n;
            PG_HIP(ctx, hipFuncSetAttribute(krows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shr));
            const int64_t units = (L + 63) / 64, blocks_r = (units + LO_WAVES - 1) / LO_WAVES, cap_r = (int64_t)ctx->cus * (direct ? RW_DIRECT_OCC : 2);
            const double *a1 = W.table, *a2 = W.tcoef;
            int32_t *a3 = W.flags;
            int64_t *a4 = W.second;
            unsigned long long *a5 = W.second_count;
            int co = (staged & 2) ? 1 : 0;
            void *args[] = {(void *)&counts_dev, &a1, &a2, &a3, &a4, &a5, (void *)&O, (void *)&P, &co};
            PG_HIP(ctx, hipLaunchKernel(krows, dim3((unsigned)(blocks_r < cap_r ? blocks_r : cap_r)), dim3(LO_THREADS), args, shr, ctx->stream));
        }
    } else {
        const double *a1 = W.table, *a2 = W.tcoef;
        int32_t *a3 = W.flags;
        int64_t *a4 = W.second;
        unsigned long long *a5 = W.second_count;
        int mm = M, st = staged;
        void *args[] = {(void *)&counts_dev, &a1, &a2, &a3, &a4, &a5, (void *)&O, (void *)&P, &mm, &st};
        PG_HIP(ctx, hipLaunchKernel(kstream, dim3(grid), dim3(LO_THREADS), args, shmem, ctx->stream));
    }
    // What the streaming pass could not close in place.  The host looks at the length of the list first (the call ends in a
    // synchronisation anyway: it has to report the complaint flag): an empty list -- clean data -- costs no launch at all, a short
    // one is taken as it is (tiles of mixed survivor counts), a long one is grouped by the number of survivors first.
    unsigned long long tail[SC_DIRTY + 1]; // (the groups, not counted yet), the complaint flag, the list's length, the dirty loci
    if (kid >= 0) pg_prof_end(ctx);
    PG_HIP(ctx, hipMemcpyAsync(tail, W.second_count, sizeof tail, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *complaint = tail[SC_COMPLAINT] != 0;
    const int64_t total = (int64_t)tail[SC_LIST];
    *listed = total;
    if ((OP == OP_OLS || OP == OP_CHISQ) && n >= 32) { // the next batch's kernel (see above): hysteresis between 0.5 % and 1 %
        if (lpl) { if ((double)tail[SC_DIRTY] < 0.005 * (double)L) rows_next = false; }
        else if ((double)total > 0.01 * (double)L) rows_next = true;
    }
    if (total == 0 || *complaint) return PG_OK;
    if (kid >= 0) pg_prof_begin(ctx, kid | PG_PROF_CONT);
    const bool p16 = ((int64_t)n * 24) % 16 == 0;
    auto pick_second = [&]() -> const void * {
        constexpr int KK = (OP == OP_OLS ? 2 : 1);
        if (OP == OP_OLS && kg == 2)
            return rns ? (p16 ? (const void *)k_locus_second<OP, 5, KK, 16> : (const void *)k_locus_second<OP, 5, KK, 8>)
                       : (p16 ? (const void *)k_locus_second<OP, 6, KK, 16> : (const void *)k_locus_second<OP, 6, KK, 8>);
        return rns ? (p16 ? (const void *)k_locus_second<OP, 5, 1, 16> : (const void *)k_locus_second<OP, 5, 1, 8>)
                   : (p16 ? (const void *)k_locus_second<OP, 6, 1, 16> : (const void *)k_locus_second<OP, 6, 1, 8>);
    };
    const void *ksecond = pick_second();
    const size_t shmem2 = (size_t)LO_WAVES * LO_TILEB + (size_t)LO_WAVES * 64 * sizeof(int64_t);
    PG_HIP(ctx, hipFuncSetAttribute(ksecond, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2));
    {
        const int64_t tiles = (total + 63) / 64 + LO_NB;
        const int64_t wants = (tiles + 3) / 4, caps = (int64_t)ctx->cus * 8;
        const unsigned gs = (unsigned)(wants < caps ? wants : caps);
        int grouped = total >= (int64_t)LO_GROUP_FROM ? 1 : 0;
        if (const char *e = std::getenv("POOLGEN_LOCUS_GROUPED")) grouped = std::atoi(e) != 0; // (A/B runs, tests of both routes)
        const int64_t *list = W.second;
        if (grouped) {
            // (few, long-running waves: every wave ends in five atomics on the same five words -- with a wave per tile they took
            // 0.34 ms per million entries, all of it contention)
            hipLaunchKernelGGL(k_locus_hist, dim3((unsigned)(wants < (int64_t)ctx->cus ? wants : (int64_t)ctx->cus)), dim3(256), 0, ctx->stream,
                               (const int64_t *)W.second, W.second_count, (const int32_t *)W.flags, total, P.pshift);
            const int64_t wantq = (tiles + 4 * SORT_CH - 1) / (4 * SORT_CH);
            hipLaunchKernelGGL(k_locus_sort, dim3((unsigned)(wantq < caps ? wantq : caps)), dim3(256), 0, ctx->stream,
                               (const int64_t *)W.second, W.second_count, (const int32_t *)W.flags, W.sorted, total, P.pshift);
            list = W.sorted;
        }
        const double *a2 = W.Y;
        int32_t *b0 = W.flags;
        double *recp = W.rec;
        const unsigned long long *b2 = W.second_count;
        int64_t tot = total;
        void *args2[] = {(void *)&counts_dev, &a2, &b0, &recp, &list, &b2, (void *)&P, &grouped, &tot};
        const int64_t want = (tiles + LO_WAVES - 1) / LO_WAVES, cap2 = (int64_t)ctx->cus * 2;
        PG_HIP(ctx, hipLaunchKernel(ksecond, dim3((unsigned)(want < cap2 ? want : cap2)), dim3(LO_THREADS), args2, shmem2, ctx->stream));
        if (OP != OP_LOAD) {
            constexpr int OPC = (OP == OP_LOAD) ? OP_OLS : OP;
            hipLaunchKernelGGL(k_locus_close<OPC>, dim3(gs), dim3(256), 0, ctx->stream,
                               (const int32_t *)W.flags, (const double *)W.rec, list,
                               (const unsigned long long *)W.second_count, (const double *)W.tcoef, O, P, grouped, total);
        }
    }
    if (kid >= 0) pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

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
    const int M = stream_period(n);
    PG_CHECK(ctx, (int64_t)64 * M * n * 24 < ((int64_t)1 << 31), "locus op: too many pools (%d) for one unit of loci", n);
    // workspace: [table n x 3 | Y n x MAXK | tcoef] [records of the second pass] [its list: L x i64] [its length: u64] [flags: i32 per locus]
    const size_t ypad = (size_t)(n + LO_CHP) * MAXK; // the second pass reads whole stages of LO_CHP pools: zeros behind the last
    const size_t side = ((size_t)n * 3 + ypad + tc.size() + 8 + 1) & ~(size_t)1; // doubles, even
    const size_t slots = (size_t)((L + 63) / 64) * 64;
    const size_t recd = slots * REC_DOUBLES;
    const size_t nsorted = (size_t)L + 64 * LO_NB;
    const size_t need = sizeof(double) * (side + recd) + sizeof(int64_t) * ((size_t)L + nsorted) + 8 * (SC_WORDS + 1) + sizeof(int32_t) * slots;
    int rc = pg_ws_reserve(ctx, need);
    if (rc) return rc;
    StreamWs W;
    W.table = static_cast<double *>(ctx->ws);
    W.Y = W.table + (size_t)n * 3;
    W.tcoef = W.Y + ypad;
    W.rec = W.table + side;
    W.second = reinterpret_cast<int64_t *>(W.rec + recd);
    W.sorted = W.second + L;
    W.second_count = reinterpret_cast<unsigned long long *>(W.sorted + nsorted);
    W.flags = reinterpret_cast<int32_t *>(W.second_count + SC_WORDS + 1);
    if (!tc.empty())
        PG_HIP(ctx, hipMemcpyAsync(W.tcoef, tc.data(), sizeof(double) * tc.size(), hipMemcpyHostToDevice, ctx->stream));
    const bool rns = flt->remove_ns != 0;
    const StreamOut O{n_out, ids, mf, stat, pv};
    // pearson_corr takes one trait per launch (its sums per trait and allele do not fit two traits at two waves per SIMD)
    const int kstep = (OP == OP_OLS) ? MAXK : 1;
    std::vector<double> Yd(ypad), tab((size_t)n * 3);
    for (int t0 = 0; t0 < k; t0 += kstep) { // the filter passes are recomputed per launch group
        const int kg = (k - t0) < kstep ? (k - t0) : kstep;
        LocusParams P;
        std::memset(&P, 0, sizeof P);
        P.L = L; P.n = n; P.k = kg; P.k_total = k; P.t0 = t0;
        P.remove_ns = flt->remove_ns ? 1 : 0;
        P.pshift = 0; // records and flags are indexed by the locus
        P.min_cov = (double)flt->min_coverage_depth;
        P.maf = flt->min_allele_frequency;
        P.max_miss = flt->max_missingness_rate;
        P.tdf = df;
        P.ntcoef = (int)tc.size();
        P.qband = 8.0 * ((double)n + 16.0) * 2.220446049250313e-16;
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
                // what the running sums over the complete pairs come to when every pair is complete (same order, same operations)
                double py = 0.0, pyy = 0.0, pn = 0.0;
                P.y_complete = 1;
                for (int i = 0; i < n; ++i) {
                    const double ye = Yd[(size_t)i * kg + t];
                    if (std::isnan(ye)) { P.y_complete = 0; continue; }
                    py = py + ye;
                    pyy = std::fma(ye, ye, pyy);
                    pn = pn + 1.0;
                }
                P.py[t] = py; P.pyy[t] = pyy; P.pn[t] = pn;
            }
        }
        const int TW = (OP == OP_OLS || OP == OP_PEARSON) ? 1 + kg : 1;
        for (int i = 0; i < n; ++i) {
            tab[(size_t)i * TW] = w[i];
            for (int t = 0; t + 1 < TW; ++t) tab[(size_t)i * TW + 1 + t] = Yd[(size_t)i * kg + t];
        }
        PG_HIP(ctx, hipMemcpyAsync(W.table, tab.data(), sizeof(double) * n * TW, hipMemcpyHostToDevice, ctx->stream));
        PG_HIP(ctx, hipMemcpyAsync(W.Y, Yd.data(), sizeof(double) * ypad, hipMemcpyHostToDevice, ctx->stream));
        int64_t listed = 0;
        bool complaint = false;
        rc = launch_passes<OP>(ctx, kid, counts_dev, W, O, P, kg, rns, &listed, &complaint); // (synchronises after the streaming pass: tab / Yd are free again)
        if (rc) return rc;
        PG_CHECK(ctx, !complaint, "locus op: a count of 2^29 (536 870 912) reads or more: beyond what the streaming pass sums exactly");
        if (t0 == 0) { ctx->lo_last_L = L; ctx->lo_last_listed = 0; }
        ctx->lo_last_listed += listed;
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

// ---- loader: filter + frequencies for every locus, one column of G per surviving allele ---------------
// FileSyncPhen::load / into_genotypes_and_phenotypes (base/sync.rs:972-1180): per locus filter
// (:195-303) -> to_frequencies over the surviving alleles (:166-192) -> with --keep-p-minus-1 sort by
// decreasing frequency and drop the first allele (:1033-1037); columns are laid out locus after locus
// in the caller's locus order.  Plan = the streaming first/second pass in OP_LOAD mode (headers only)
// + an exclusive scan of the column counts; emit = one thread per (locus, pool) that re-reads its six
// counts and writes its frequencies, consecutive threads to consecutive doubles of a G row.
__device__ __forceinline__ int load_ncols(int hdr, int kpm1) {
    const int nk = (hdr >> H_NK_SHIFT) & 7;
    const int c = (hdr & FLAG_ALIVE) ? nk - kpm1 : 0;
    return c > 0 ? c : 0;
}

__global__ __launch_bounds__(256) void k_load_count(const int32_t *__restrict__ flags, const int64_t *__restrict__ order,
                                                    int64_t L, int pshift, int kpm1, int32_t *__restrict__ local,
                                                    int64_t *__restrict__ blocksum) {
    __shared__ int sc[256];
    const int tid = threadIdx.x;
    const int64_t oi = (int64_t)blockIdx.x * 256 + tid;
    int c = 0;
    if (oi < L) c = load_ncols(flags[unit_slot(order ? order[oi] : oi, pshift)], kpm1);
    sc[tid] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? sc[tid - d] : 0;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    if (oi < L) local[oi] = sc[tid] - c;
    if (tid == 255) blocksum[blockIdx.x] = sc[255];
}

__global__ __launch_bounds__(1024) void k_load_scan(const int64_t *__restrict__ blocksum, int64_t nb,
                                                    int64_t *__restrict__ blockoff, int64_t *__restrict__ total) {
    __shared__ int64_t sc[1024];
    __shared__ int64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + tid;
        const int64_t c = i < nb ? blocksum[i] : 0;
        sc[tid] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int64_t v = tid >= d ? sc[tid - d] : 0;
            __syncthreads();
            sc[tid] += v;
            __syncthreads();
        }
        if (i < nb) blockoff[i] = carry + sc[tid] - c;
        __syncthreads();
        if (tid == 1023) carry += sc[1023];
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_load_emit(const uint32_t *__restrict__ counts, const int32_t *__restrict__ flags,
                                                   const int64_t *__restrict__ order, const int32_t *__restrict__ local,
                                                   const int64_t *__restrict__ blockoff, int64_t L, int n, int pshift,
                                                   int kpm1, const int32_t *__restrict__ pool_map, int n_out,
                                                   double *__restrict__ G, int64_t ld, int64_t *__restrict__ col_locus,
                                                   int32_t *__restrict__ col_allele, double *__restrict__ cov) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t oi = gid / n;
    if (oi >= L) return;
    const int pool = (int)(gid - oi * n);
    const int64_t l = order ? order[oi] : oi;
    const int hdr = flags[unit_slot(l, pshift)];
    const int ncols = load_ncols(hdr, kpm1);
    if (ncols == 0) return;
    const int64_t off = blockoff[oi >> 8] + local[oi];
    const uint2_t *cp = reinterpret_cast<const uint2_t *>(counts + ((size_t)l * n + pool) * 6);
    const uint2_t a = cp[0], b = cp[1], d = cp[2];
    const uint32_t c[NA] = {a.x, a.y, b.x, b.y, d.x, d.y};
    double rs = 0.0; // row sum over the surviving alleles in column order (sync.rs:170-175)
#pragma unroll
    for (int j = 0; j < NA; ++j) rs = (hdr & (2 << j)) ? rs + (double)c[j] : rs;
    const int ordbits = hdr >> H_ORD_SHIFT;
    const int po = pool_map ? pool_map[pool] : pool;
    for (int r = 0; r < ncols; ++r) {
        const int al = (ordbits >> (3 * (r + kpm1))) & 7;
        uint32_t cv = c[0];
#pragma unroll
        for (int j = 1; j < NA; ++j) cv = (al == j) ? c[j] : cv;
        const double f = (rs == 0.0) ? NAN : (double)cv / rs; // sync.rs:176-183
        double *row = G + (size_t)(off + r) * ld;
        if (po >= 0) row[po] = f;
        if (cov && po >= 0) cov[(size_t)(off + r) * ld + po] = rs; // the pool's depth over the surviving alleles (sync.rs:1142-1152)
        if (pool == 0) {
            col_locus[off + r] = l;
            col_allele[off + r] = al;
            for (int64_t q = n_out; q < ld; ++q) row[q] = 0.0; // padding columns of the locus-major layout
            if (cov)
                for (int64_t q = n_out; q < ld; ++q) cov[(size_t)(off + r) * ld + q] = 0.0;
        }
    }
}

int load_plan(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n, const double *pool_sizes,
              const pg_filter *flt, int keep_p_minus_1, const int64_t *order_dev, int64_t *p_out) {
    PG_CHECK(ctx, counts_dev && pool_sizes && flt && p_out, "load: null pointer");
    PG_CHECK(ctx, L > 0 && n >= 1, "load: bad shape L=%lld n=%d", (long long)L, n);
    PG_CHECK(ctx, (reinterpret_cast<uintptr_t>(counts_dev) & 15) == 0, "load: counts must be 16-byte aligned");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> w(n);
    double total = 0.0;
    for (int i = 0; i < n; ++i) total = total + pool_sizes[i];
    for (int i = 0; i < n; ++i) w[i] = pool_sizes[i] / total; // sync.rs:266-268
    const int M = stream_period(n);
    PG_CHECK(ctx, (int64_t)64 * M * n * 24 < ((int64_t)1 << 31), "load: too many pools (%d) for one unit of loci", n);
    const size_t slots = (size_t)((L + 63) / 64) * 64;
    const int64_t nb = (L + 255) / 256;
    // workspace: [w][flags: i32 per locus][second list: L x i64][its length][local: L x i32][blocksum][blockoff][total][pool map]
    auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off = 0;
    const size_t o_w = off; off = al16(off + sizeof(double) * n);
    const size_t o_flags = off; off = al16(off + sizeof(int32_t) * slots);
    const size_t o_second = off; off = al16(off + sizeof(int64_t) * (size_t)L);
    const size_t o_sorted = off; off = al16(off + sizeof(int64_t) * ((size_t)L + 64 * LO_NB));
    const size_t o_count = off; off = al16(off + 8 * (SC_WORDS + 1));
    const size_t o_local = off; off = al16(off + sizeof(int32_t) * (size_t)L);
    const size_t o_bsum = off; off = al16(off + 8 * (size_t)nb);
    const size_t o_boff = off; off = al16(off + 8 * (size_t)nb);
    const size_t o_total = off; off = al16(off + 8);
    const size_t o_pmap = off; off = al16(off + sizeof(int32_t) * (size_t)n);
    int rc = pg_ws_reserve(ctx, off);
    if (rc) return rc;
    char *ws = static_cast<char *>(ctx->ws);
    StreamWs W;
    W.table = reinterpret_cast<double *>(ws + o_w);
    W.Y = nullptr; W.tcoef = nullptr; W.rec = nullptr;
    W.flags = reinterpret_cast<int32_t *>(ws + o_flags);
    W.second = reinterpret_cast<int64_t *>(ws + o_second);
    W.sorted = reinterpret_cast<int64_t *>(ws + o_sorted);
    W.second_count = reinterpret_cast<unsigned long long *>(ws + o_count);
    int32_t *recf = W.flags;
    int32_t *local = reinterpret_cast<int32_t *>(ws + o_local);
    int64_t *bsum = reinterpret_cast<int64_t *>(ws + o_bsum), *boff = reinterpret_cast<int64_t *>(ws + o_boff);
    int64_t *tot_dev = reinterpret_cast<int64_t *>(ws + o_total);
    PG_HIP(ctx, hipMemcpyAsync(W.table, w.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    LocusParams P;
    std::memset(&P, 0, sizeof P);
    P.L = L; P.n = n; P.k = 1; P.k_total = 1; P.t0 = 0;
    P.remove_ns = flt->remove_ns ? 1 : 0;
    P.pshift = 0; // flags are indexed by the locus
    P.sort_desc = keep_p_minus_1 ? 1 : 0;
    P.min_cov = (double)flt->min_coverage_depth;
    P.maf = flt->min_allele_frequency;
    P.max_miss = flt->max_missingness_rate;
    P.qband = 8.0 * ((double)n + 16.0) * 2.220446049250313e-16;
    P.y_complete = 1;
    int64_t listed = 0;
    bool complaint = false;
    rc = launch_passes<OP_LOAD>(ctx, -1, counts_dev, W, StreamOut{nullptr, nullptr, nullptr, nullptr, nullptr}, P, 1, flt->remove_ns != 0,
                                &listed, &complaint);
    if (rc) return rc;
    PG_CHECK(ctx, !complaint, "load: a count of 2^29 (536 870 912) reads or more: beyond what the streaming pass sums exactly");
    hipLaunchKernelGGL(k_load_count, dim3((unsigned)nb), dim3(256), 0, ctx->stream, recf, order_dev, L, P.pshift,
                       P.sort_desc, local, bsum);
    hipLaunchKernelGGL(k_load_scan, dim3(1), dim3(1024), 0, ctx->stream, bsum, nb, boff, tot_dev);
    PG_HIP(ctx, hipGetLastError());
    int64_t tot = 0;
    PG_HIP(ctx, hipMemcpyAsync(&tot, tot_dev, 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->lo_last_L = L; ctx->lo_last_listed = listed;
    *p_out = tot;
    ctx->load_valid = true;
    ctx->load_counts = counts_dev; ctx->load_order = order_dev;
    ctx->load_L = L; ctx->load_total = tot; ctx->load_nunits = 0;
    ctx->load_n = n; ctx->load_kpm1 = P.sort_desc; ctx->load_pshift = P.pshift;
    ctx->load_off_flags = o_flags; ctx->load_off_local = o_local; ctx->load_off_blockoff = o_boff;
    ctx->load_off_poolmap = o_pmap;
    return PG_OK;
}

int load_emit(pg_ctx *ctx, const int32_t *pool_map, int n_out, double *G_dev, int64_t ld, int64_t *col_locus_dev,
              int32_t *col_allele_dev, double *cov_dev) {
    if (!ctx->load_valid) return pg_fail(ctx, PG_ERR_STATE, "load_emit: call pg_load_plan_dev first (and nothing else in between)");
    PG_CHECK(ctx, G_dev && col_locus_dev && col_allele_dev, "load_emit: null pointer");
    const int n = ctx->load_n;
    if (!pool_map) n_out = n;
    PG_CHECK(ctx, n_out >= 1 && ld >= n_out && (ld % 2) == 0, "load_emit: ld (%lld) must be even and >= the pools kept (%d)",
             (long long)ld, n_out);
    if (pool_map)
        for (int i = 0; i < n; ++i) PG_CHECK(ctx, pool_map[i] >= -1 && pool_map[i] < n_out, "load_emit: pool_map[%d] out of range", i);
    if (ctx->load_total == 0) return PG_OK;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    char *ws = static_cast<char *>(ctx->ws);
    int32_t *pmap_dev = nullptr;
    if (pool_map) {
        pmap_dev = reinterpret_cast<int32_t *>(ws + ctx->load_off_poolmap);
        PG_HIP(ctx, hipMemcpyAsync(pmap_dev, pool_map, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    }
    const int64_t threads = ctx->load_L * n;
    hipLaunchKernelGGL(k_load_emit, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, ctx->load_counts,
                       reinterpret_cast<const int32_t *>(ws + ctx->load_off_flags), ctx->load_order,
                       reinterpret_cast<const int32_t *>(ws + ctx->load_off_local),
                       reinterpret_cast<const int64_t *>(ws + ctx->load_off_blockoff), ctx->load_L, n, ctx->load_pshift,
                       ctx->load_kpm1, pmap_dev, n_out, G_dev, ld, col_locus_dev, col_allele_dev, cov_dev);
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // pool_map is the caller's
    return PG_OK;
}

} // namespace

extern "C" int pg_locus_op_stats(const pg_ctx *ctx, int64_t *loci, int64_t *listed) {
    if (!ctx) return PG_ERR_INVALID;
    if (loci) *loci = ctx->lo_last_L;
    if (listed) *listed = ctx->lo_last_listed;
    return PG_OK;
}

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

namespace {
// 16-bit counts (what the host parser stores when every count fits) -> the 32-bit device layout; 8 values per thread
__global__ __launch_bounds__(256) void k_expand_u16(const uint4_t *__restrict__ src, uint4_t *__restrict__ dst, int64_t n8,
                                                    const uint16_t *__restrict__ src_tail, uint32_t *__restrict__ dst_tail,
                                                    int ntail) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n8) {
        const uint4_t v = src[i];
        uint4_t a, b;
        a.x = v.x & 0xFFFFu; a.y = v.x >> 16; a.z = v.y & 0xFFFFu; a.w = v.y >> 16;
        b.x = v.z & 0xFFFFu; b.y = v.z >> 16; b.z = v.w & 0xFFFFu; b.w = v.w >> 16;
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
    if (i == 0)
        for (int t = 0; t < ntail; ++t) dst_tail[t] = src_tail[t];
}
} // namespace

extern "C" int pg_expand_counts_u16_dev(pg_ctx *ctx, const uint16_t *src_dev, int64_t n_values, uint32_t *dst_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, src_dev && dst_dev && n_values >= 0, "expand_counts: bad arguments");
    PG_CHECK(ctx, (reinterpret_cast<uintptr_t>(src_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_dev) & 15) == 0,
             "expand_counts: buffers must be 16-byte aligned");
    if (n_values == 0) return PG_OK;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n8 = n_values / 8;
    const int ntail = (int)(n_values - 8 * n8);
    const int64_t blocks = std::max<int64_t>(1, (n8 + 255) / 256);
    hipLaunchKernelGGL(k_expand_u16, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4_t *>(src_dev),
                       reinterpret_cast<uint4_t *>(dst_dev), n8, src_dev + 8 * n8, dst_dev + 8 * n8, ntail);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

extern "C" int pg_load_plan_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n, const double *pool_sizes,
                                const pg_filter *filter, int keep_p_minus_1, const int64_t *order_dev, int64_t *p_out) {
    if (!ctx) return PG_ERR_INVALID;
    return load_plan(ctx, counts_dev, L, n, pool_sizes, filter, keep_p_minus_1, order_dev, p_out);
}

extern "C" int pg_load_emit_dev(pg_ctx *ctx, const int32_t *pool_map, int n_out, double *G_dev, int64_t ld,
                                int64_t *col_locus_dev, int32_t *col_allele_dev) {
    if (!ctx) return PG_ERR_INVALID;
    return load_emit(ctx, pool_map, n_out, G_dev, ld, col_locus_dev, col_allele_dev, nullptr);
}

extern "C" int pg_load_emit_cov_dev(pg_ctx *ctx, const int32_t *pool_map, int n_out, double *G_dev, int64_t ld,
                                    int64_t *col_locus_dev, int32_t *col_allele_dev, double *cov_dev) {
    if (!ctx) return PG_ERR_INVALID;
    return load_emit(ctx, pool_map, n_out, G_dev, ld, col_locus_dev, col_allele_dev, cov_dev);
}
