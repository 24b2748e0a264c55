// pg_kinship.hip -- S = sum_l g_l g_l^T over a slab of loci, fp64 MFMA (v_mfma_f64_16x16x4_f64).
//
// Reference: `kinship = g.dot(&g.t()) / p` (gwas/ols.rs:291-295) and the X X^T contraction of
// gp::ols (gp/ols.rs:56, base/helpers.rs:222-255).  G is locus-major (p x n, ld), so both MFMA
// operands of a 16x16 output tile are "4 loci x 16 pools" fragments of the SAME rows:
//     A[i][k] = G[l0+k][a0+i]   (lane l: i = l&15, k = l>>4)
//     B[k][j] = G[l0+k][b0+j]   (lane l: j = l&15, k = l>>4)
// i.e. 16 consecutive lanes read 128 contiguous bytes of one locus row -- no transposition.
//
// Decomposition.  One 1024-thread workgroup (16 waves, 4 per SIMD) owns a contiguous slab of
// loci and a pair (bi <= bj) of pool blocks (a block = up to `Tb` 16-pool tiles).  For
// n <= 208 pools there is a single block: the workgroup holds the WHOLE upper triangle of S
// (<= 91 tiles, <= 6 per wave, 4 fp64 accumulators per lane per tile) in registers while it
// streams its loci once from HBM through a double-buffered LDS stage (16 loci per stage).
// The contraction is MFMA-bound (intensity n/4 flop per byte); HBM traffic is the single read
// of G.  Each workgroup finally writes its partial tiles to a slab; a second tiny kernel sums
// the slabs in a fixed order (deterministic, no atomics) and mirrors the lower triangle.
#include "pg_common.h"
#include <cstdlib>
#include <cstring>
#include <utility>

namespace {

#ifndef KIN_WAVES_DEF
#define KIN_WAVES_DEF 16
#endif
constexpr int KIN_WAVES = KIN_WAVES_DEF; // 16 (6 tile slots per wave) or 8 (12 slots)
constexpr int KIN_THREADS = 64 * KIN_WAVES;
#ifndef KIN_RING_D
#define KIN_RING_D 3
#endif
constexpr int KIN_FUSE_MAXK = 2; // traits the fused intercept-only pass carries
constexpr int KIN_TPW = 96 / KIN_WAVES;  // max tiles per wave (>= ceil(91 / waves), ceil(81 / waves))
#ifndef KIN_KC_DEF
#define KIN_KC_DEF 16
#endif
constexpr int KIN_KC = KIN_KC_DEF;  // loci per LDS stage (4 loci = one MFMA k-step)
constexpr int KIN_PPT = (KIN_KC * 128 + KIN_THREADS - 1) / KIN_THREADS;  // 16-byte staging pieces per thread per stage

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

// v + (v shifted by a DPP control) on the VALU only: the generic __shfl_xor goes through ds_bpermute,
// whose latency would stall the issuing wave in the middle of the MFMA stream.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
// value of lane ((l & 15) + N) % 16 of the same 16-lane row (row_ror:(16 - N) moves data towards higher lanes by 16 - N)
template <int N>
__device__ __forceinline__ double row_from_plus(double v) {
    static_assert(N >= 1 && N <= 15, "row rotation");
    constexpr int CTRL = 0x120 + (16 - N); // row_ror:(16 - N)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// ---- compile-time helpers ------------------------------------------------------------------------
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// row-major enumeration of the upper triangle of a T x T tile grid
constexpr int kin_tri_ti(int t, int T) { int ti = 0; while (ti < T && t >= T - ti) { t -= T - ti; ++ti; } return ti < T ? ti : 0; }
constexpr int kin_tri_tj(int t, int T) { int ti = 0; while (ti < T && t >= T - ti) { t -= T - ti; ++ti; } return ti < T ? ti + t : 0; }
// tile of slot u of wave w (slots past the end of the list recompute tile 0 and are never stored)
constexpr int kin_slot_tile(int w, int u, int T) { return (w + KIN_WAVES_DEF * u) < T * (T + 1) / 2 ? (w + KIN_WAVES_DEF * u) : 0; }

// ---- 13-tile shape: which tiles a wave owns, chosen so that they SHARE fragments ------------------------------
// A fragment "4 loci x 16 pools of tile column c" serves as the A operand of every tile in tile row c and as the B
// operand of every tile in tile column c.  Waves 0..3 own the 3 x 3 triangles on the diagonal (6 tiles from 3
// fragments), waves 4..9 the 2 x 3 blocks (6 tiles from 5 fragments), the rest what is left, grouped for shared
// columns: 69 fragment reads per k-step for the 91 tiles instead of 182, and 23/23/23/22 tiles on the four SIMDs.
constexpr int KIN13_T[16][6][2] = {
    {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}},
    {{3, 3}, {3, 4}, {3, 5}, {4, 4}, {4, 5}, {5, 5}},
    {{6, 6}, {6, 7}, {6, 8}, {7, 7}, {7, 8}, {8, 8}},
    {{9, 9}, {9, 10}, {9, 11}, {10, 10}, {10, 11}, {11, 11}},
    {{0, 3}, {0, 4}, {0, 5}, {1, 3}, {1, 4}, {1, 5}},
    {{0, 6}, {0, 7}, {0, 8}, {1, 6}, {1, 7}, {1, 8}},
    {{0, 9}, {0, 10}, {0, 11}, {1, 9}, {1, 10}, {1, 11}},
    {{3, 6}, {3, 7}, {3, 8}, {4, 6}, {4, 7}, {4, 8}},
    {{3, 9}, {3, 10}, {3, 11}, {4, 9}, {4, 10}, {4, 11}},
    {{6, 9}, {6, 10}, {6, 11}, {7, 9}, {7, 10}, {7, 11}},
    {{2, 4}, {2, 11}, {2, 12}, {4, 12}, {11, 12}, {12, 12}},
    {{2, 5}, {2, 6}, {2, 7}, {5, 6}, {5, 7}, {-1, -1}},
    // (rows 12, 13, 15 in this order so that the four waves of a SIMD -- w, w + 4, w + 8, w + 12 -- carry nearly equal matrix time
    //  once the narrow tiles run on 4 x 4 x 4 blocks: 1339 / 1369 / 1357 / 1334 cycles per 4 loci)
    {{0, 12}, {2, 3}, {3, 12}, {6, 12}, {7, 12}, {-1, -1}},
    {{5, 9}, {5, 10}, {5, 12}, {9, 12}, {10, 12}, {-1, -1}},
    {{2, 8}, {2, 9}, {2, 10}, {8, 9}, {8, 10}, {-1, -1}},
    {{1, 12}, {5, 8}, {5, 11}, {8, 11}, {8, 12}, {-1, -1}},
};
constexpr int kin13_ntiles(int w) { int c = 0; for (int u = 0; u < 6; ++u) c += KIN13_T[w][u][0] >= 0; return c; }
// distinct tile columns of wave w, in order of first use; kin13_col(w, i) = -1 past the end
constexpr int kin13_col(int w, int idx) {
    int seen[12] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, ns = 0;
    for (int u = 0; u < 6; ++u)
        for (int h = 0; h < 2; ++h) {
            const int c = KIN13_T[w][u][h];
            if (c < 0) continue;
            bool dup = false;
            for (int i = 0; i < ns; ++i) dup = dup || seen[i] == c;
            if (!dup) seen[ns++] = c;
        }
    return idx < ns ? seen[idx] : -1;
}
constexpr int kin13_ncols(int w) { int c = 0; while (c < 12 && kin13_col(w, c) >= 0) ++c; return c; }
constexpr int kin13_slot(int w, int col) { int i = 0; while (kin13_col(w, i) != col) ++i; return i; } // fragment register of a column
constexpr int KIN13_MAXC = 6;
constexpr int kin13_ndiag(int w) { int c = 0; for (int u = 0; u < 6; ++u) c += (KIN13_T[w][u][0] >= 0 && KIN13_T[w][u][0] == KIN13_T[w][u][1]); return c; }
constexpr int kin13_diag_index(int w, int u) { int c = 0; for (int v = 0; v < u; ++v) c += (KIN13_T[w][v][0] >= 0 && KIN13_T[w][v][0] == KIN13_T[w][v][1]); return c; }

struct KinParams {
    const double *G;
    int64_t p;
    int64_t ld;
    int64_t loci_per_wg;
    double *slabs; // [gridDim.x][npad * npad]
    int n, T, npad, Tb, nb;
    int split;  // nb == 2: the one pair's tile list (A x B, A x A, B x B: 136 tiles at T = 16) is dealt to `split` = 2
                // workgroups per slab of loci instead of three workgroups of which two are a third full
    int merged; // nb >= 4: no workgroups for the diagonal blocks -- every off-diagonal pair (A, B) also takes 1/(nb-1) of
                // the tiles of A x A and of B x B (64 + 12 + 12 of its 96 tile slots at Tb = 8 instead of 64, or 36 on a
                // diagonal workgroup), and every pool column is staged nb - 1 times instead of nb
    // fused speculative intercept-only sums (FUSE kernels only)
    const double *ytil; // k x 256 centred phenotypes, zero padded
    double *spec;       // p x (2 + k)
    int k;
    // More than one pool-block pair (n > 208): the workgroups that stream the SAME slab of loci are placed on ONE XCD, so that
    // a slab's lines reach HBM once and the other pairs that stage them hit that XCD's L2 (workgroups go to the XCDs round-robin
    // by linear id; with (slab, pair) = blockIdx.(x, y) the pairs of a slab landed on different XCDs and each pool column came
    // from HBM nb - 1 times: 54 GB for the 20 GB of config 4).  xcd_spx = slabs per XCD in that placement, 0 = plain 2-D grid.
    int xcd_spx, npairs, nslab;
    // run-time tile tables: a wave skips the slots it has no tile for (28 tiles on 16 waves: 4 of 32 slots; 10 tiles on 8 waves: 6 of
    // 16) -- their LDS reads and matrix cycles.  4 M loci, ms, without -> with: 32 pools 0.89 -> 0.77, 64: 1.02 -> 0.95, 100: 1.54 ->
    // 1.43, 150: 2.57 -> 2.36, 250 (merged pair lists): 6.83 -> 5.53.  Off for the weighted pairs (wq_n), whose slab lengths are
    // balanced on the slot counts (224 and 300 pools: 10-15 % slower with it), and from four blocks on.
    int skip_dead; // (host side only: selects the kernel instance)
    // Two or three pool blocks without the merged tile lists (n = 209..224, 257..384): the pairs differ in work (a diagonal pair
    // of 8 tile columns has 36 tiles, an off-diagonal one 64, the pairs of a short last block fewer still), so each pair gets its
    // own number of workgroups -- and with it its own slab length -- in proportion to the tile slots its waves run.  With one
    // slab length for all, the diagonal workgroups finished early and a third of the CUs idled (n = 300: 47 of 79 TFLOP/s).
    // wq_n[q] > 0 switches this on: pair q = workgroups wq_first[q] .. + wq_n[q] of a 1-D grid, wq_len[q] loci each.
    int wq_n[6], wq_first[6];
    int64_t wq_len[6];
};

// SMALL (13-tile shape only): the tiles that are not full 16 x 16 blocks of useful products run on v_mfma_f64_4x4x4_4b_f64
// (four independent 4 x 4 x 4 blocks per instruction, a quarter of the matrix-pipe time of a 16 x 16 x 4):
//   * the 13 diagonal tiles need their upper triangle only: 3 instructions (the block diagonals d = 0, 1, 2 of the 4 x 4 grid of
//     4 x 4 blocks; the wrapped block of d = 1 is the transpose of the one block d = 3 would add) instead of 4 instructions' worth;
//   * with n <= 200 pools tile column 12 holds at most 8 pools: 2 instructions (4 row blocks x one 4-pool column block each)
//     per tile instead of 4.
// Lane layout of the instruction (probed on the device, tools/probe_mfma4.hip): A[blk][i][k] in lane 16 k + 4 blk + i,
// B[blk][k][j] in lane 16 k + 4 blk + j, D[blk][i][j] in lane 16 i + 4 blk + j -- so the 16 x 16 fragment of tile column c
// (lane: pool = lane & 15, locus = lane >> 4) IS the A operand whose four blocks are the four 4-pool groups of that column,
// a B operand "column group (blk + d) % 4" is the same register rotated by 4 d lanes inside its 16-lane row (DPP, no LDS), and
// only the 8-pool strip needs reads of its own.  200 pools: 5 200 instead of 5 824 matrix cycles per 4 loci.
// SKIP (run-time tile tables only): see KinParams::skip_dead -- a template parameter because even a never-taken branch around an
// item keeps the compiler from running the fragment reads ahead of the MFMAs (measured: +8..10 % at 224 and 500 pools).
template <bool FUSE, bool SPEC13, int SMALL = 0, bool SKIP = false> // SMALL bits: 1 = diagonal tiles on 4 x 4 blocks, 2 = 8-pool last column (n <= 200) on 4 x 4 blocks
__global__ __launch_bounds__(KIN_THREADS, 1) void k_kinship_syrk(KinParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fi = lane & 15; // pool within fragment / output column
    const int kq = lane >> 4; // locus within k-step / output row group

    // ---- which slab of loci, which pair of pool blocks ------------------------------------
    int bx = blockIdx.x, by = blockIdx.y;
    if (P.xcd_spx) { // 1-D grid: linear id -> (XCD, position on it) -> (slab, pair); see KinParams::xcd_spx
        const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3, main = P.xcd_spx * P.npairs;
        if (s < main) { bx = xcd * P.xcd_spx + s / P.npairs; by = s % P.npairs; }
        else { const int e = (s - main) * 8 + xcd; bx = 8 * P.xcd_spx + e / P.npairs; by = e % P.npairs; } // the slabs left over
        if (bx >= P.nslab) return;
    }
    int64_t loci_per_wg = P.loci_per_wg;
    if (P.wq_n[0] > 0) { // weighted pairs: see KinParams::wq_n
        int q = 0;
        while (q + 1 < P.npairs && (int)blockIdx.x >= P.wq_first[q + 1]) ++q;
        by = q;
        bx = (int)blockIdx.x - P.wq_first[q];
        loci_per_wg = P.wq_len[q];
    }
    int bi = 0, bj = 0;
    {
        int q = by / P.split;
        const int first = P.merged ? 1 : 0; // merged: pairs with bi < bj only
        for (bi = 0; bi < P.nb; ++bi) {
            const int cnt = P.nb - bi - first;
            if (q < cnt) { bj = bi + first + q; break; }
            q -= cnt;
        }
    }
    const bool diag = (bi == bj);
    const int Ta = min(P.Tb, P.T - bi * P.Tb);
    const int Tbb = min(P.Tb, P.T - bj * P.Tb);
    const int colsA = Ta * 16;
    const int colsB = diag ? 0 : Tbb * 16;
    int ldsld = colsA + colsB;
    if ((ldsld & 31) != 16) ldsld += 16; // rows 2 apart hit disjoint bank halves (ds_read_b64)
    const int a0 = bi * P.Tb * 16;
    const int b0 = bj * P.Tb * 16;
    const int wa = min(colsA, P.n - a0);
    const int wb = diag ? 0 : min(colsB, P.n - b0);
    const int npa = (wa + 1) >> 1, npb = (wb + 1) >> 1;
    const int npr = npa + npb; // 16-byte pieces per staged locus row

    // ---- this wave's tiles ---------------------------------------------------------------
    const int nrect = diag ? 0 : Ta * Tbb;
    const int NA = Ta * (Ta + 1) / 2, NB = Tbb * (Tbb + 1) / 2, part = P.nb - 1;
    const int rankA = bj - 1, rankB = bi; // position of the partner among the other blocks, in block order
    const int cntA = P.merged ? (NA > rankA ? (NA - rankA + part - 1) / part : 0) : 0;
    const int cntB = P.merged ? (NB > rankB ? (NB - rankB + part - 1) / part : 0) : 0;
    const int ntiles = diag ? NA : nrect + cntA + cntB;
    const int split_id = by % P.split;
    // Every wave runs exactly KIN_TPW tile slots so that the k-loop is straight-line code; a slot
    // beyond the tile list recomputes tile 0 into an accumulator that is never stored.
    int acol[KIN_TPW], bcol[KIN_TPW], orow[KIN_TPW], ocol[KIN_TPW];
    bool live[KIN_TPW];
#pragma unroll
    for (int u = 0; u < KIN_TPW; ++u) {
        const int t = split_id + P.split * (wave + KIN_WAVES * u);
        int ti = 0, tj = 0;
        live[u] = t < ntiles;
        int kind = diag ? 1 : 0; // 0: A x B, 1: inside A, 2: inside B
        if (t < ntiles) {
            int q = t, Td = Ta;
            if (!diag && t >= nrect) {
                const bool inA = t - nrect < cntA;
                kind = inA ? 1 : 2;
                q = inA ? rankA + part * (t - nrect) : rankB + part * (t - nrect - cntA);
                Td = inA ? Ta : Tbb;
            }
            if (kind) {
                for (ti = 0; ti < Td; ++ti) {
                    const int cnt = Td - ti;
                    if (q < cnt) { tj = ti + q; break; }
                    q -= cnt;
                }
            } else {
                ti = t / Tbb;
                tj = t - ti * Tbb;
            }
        }
        acol[u] = (kind == 2 ? colsA : 0) + 16 * ti;
        bcol[u] = (kind == 1 ? 0 : colsA) + 16 * tj;
        orow[u] = (kind == 2 ? b0 : a0) + 16 * ti;
        ocol[u] = (kind == 1 ? a0 : b0) + 16 * tj;
    }

    // ---- staging assignment (constant over the stages) -------------------------------------
    int st_loc[KIN_PPT], st_lcol[KIN_PPT];
    int st_gcol[KIN_PPT];
    bool st_on[KIN_PPT], st_two[KIN_PPT];
#pragma unroll
    for (int r = 0; r < KIN_PPT; ++r) {
        const int t = tid + r * KIN_THREADS;
        const int loc = t / npr;
        const int q = t - loc * npr;
        st_on[r] = loc < KIN_KC;
        st_loc[r] = loc;
        int gcol, lcol;
        bool two;
        if (q < npa) {
            gcol = a0 + 2 * q; lcol = 2 * q; two = (2 * q + 1) < wa;
        } else {
            const int q2 = q - npa;
            gcol = b0 + 2 * q2; lcol = colsA + 2 * q2; two = (2 * q2 + 1) < wb;
        }
        st_lcol[r] = lcol;
        st_two[r] = two;
        st_gcol[r] = st_on[r] ? gcol : a0;
        if (!st_on[r]) st_loc[r] = 0;
    }

    const int64_t l_begin = (int64_t)bx * loci_per_wg;
    const int64_t l_end = min(P.p, l_begin + loci_per_wg);
    const int nstages = (l_end > l_begin) ? (int)((l_end - l_begin + KIN_KC - 1) / KIN_KC) : 0;
    const int bufsz = KIN_KC * ldsld;

    // zero both buffers once: the padding columns must stay zero for the edge tiles
    for (int i = tid; i < 2 * bufsz; i += KIN_THREADS) lds[i] = 0.0;
    double *ylds = lds + 2 * bufsz; // FUSE: KIN_FUSE_MAXK x 256 centred phenotypes, then the partial-sum scratch
    if (FUSE)
        for (int i = tid; i < P.k * 256; i += KIN_THREADS) ylds[i] = P.ytil[i];
    __syncthreads();

    // Staging uses raw buffer loads over a descriptor that covers exactly this workgroup's slab:
    // pieces of loci past the slab end are out of range and come back as zeros in hardware, so
    // the loop carries no tail branches and the loads stay in flight across the MFMA block.
    const uint64_t slab_bytes64 = (uint64_t)(l_end > l_begin ? l_end - l_begin : 0) * P.ld * 8;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(P.G + l_begin * P.ld), 0, (int)(uint32_t)slab_bytes64, 0x00020000);
    uint4_t stage_reg[KIN_PPT];
    uint32_t st_voff[KIN_PPT];
#pragma unroll
    for (int r = 0; r < KIN_PPT; ++r)
        st_voff[r] = st_on[r] ? (uint32_t)(((int64_t)st_loc[r] * P.ld + st_gcol[r]) * 8) : 0xFFFFFFF0u;
    const uint32_t stage_stride = (uint32_t)(KIN_KC * P.ld * 8);
    auto stage_load = [&](int c) {
#pragma unroll
        for (int r = 0; r < KIN_PPT; ++r)
            // the whole offset goes through VGPR: the scalar offset is not range-checked on gfx9
            stage_reg[r] = __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, st_on[r] ? st_voff[r] + (uint32_t)c * stage_stride : 0xFFFFFFF0u, 0, 0);
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int r = 0; r < KIN_PPT; ++r) {
            if (st_on[r]) {
                uint4_t v = stage_reg[r];
                if (!st_two[r]) { v.z = 0u; v.w = 0u; }
                *reinterpret_cast<uint4_t *>(&lds[buf * bufsz + st_loc[r] * ldsld + st_lcol[r]]) = v;
            }
        }
    };

    if (nstages > 0) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();

    // FUSE: from the stage that is in LDS anyway, wave w also forms the sums an intercept-only fit of
    // locus w of the stage needs (g' = g - g[0]):  sum g', sum g'^2, sum g' ytil_t.  Its 64 lanes cover
    // the pools 4 apiece; the 64 partials per value are transposed through a wave-private LDS scratch
    // so that row v of the wave (16 lanes) holds value v, 4 partials per lane, and a 4-step intra-row
    // DPP shift-add finishes the sum in lanes 15/31/47/63 -- no ds_bpermute, no cross-row step, no
    // extra barrier (a wave's own LDS writes and reads are ordered).
    const int NV = 2 + P.k;
    double *scratch = ylds + KIN_FUSE_MAXK * 256 + wave * (4 * 64); // [4 values][64 lanes]
    // this lane's four pools' centred phenotypes: registers, not one more LDS read per use
    double yreg[KIN_FUSE_MAXK][4];
#pragma unroll
    for (int t = 0; t < KIN_FUSE_MAXK; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) yreg[t][j] = (FUSE && t < P.k) ? ylds[t * 256 + lane + 64 * j] : 0.0;
    auto spec_pass = [&](const double *buf, int64_t lbase) {
#pragma unroll
        for (int jl = 0; jl < KIN_KC / KIN_WAVES; ++jl) {
            const int loc = wave + KIN_WAVES * jl;
            const double *row = buf + loc * ldsld;
            const double shift = row[0]; // same address in every lane: an LDS broadcast
            double s1 = 0.0, s2 = 0.0, sy[KIN_FUSE_MAXK], d[4];
#pragma unroll
            for (int t = 0; t < KIN_FUSE_MAXK; ++t) sy[t] = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pool = lane + 64 * j;
                // 13-tile shape: 193..208 pools, pitch 208 -- the first three 64-pool groups are always whole
                if (SPEC13 && j < 3) d[j] = row[pool] - shift;
                else {
                    const double gv = row[pool < ldsld ? pool : ldsld - 1];
                    d[j] = (pool < P.n) ? gv - shift : 0.0;
                }
                s1 += d[j];
                s2 = fma(d[j], d[j], s2);
                sy[0] = fma(d[j], yreg[0][j], sy[0]);
            }
            const bool two = P.k > 1; // wave-uniform
            if (two) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sy[1] = fma(d[j], yreg[1][j], sy[1]);
            }
            scratch[lane] = s1;
            scratch[64 + lane] = s2;
            scratch[2 * 64 + lane] = sy[0];
            if (two) scratch[3 * 64 + lane] = sy[1];
            __builtin_amdgcn_wave_barrier();
            const int v = lane >> 4, r = lane & 15;
            const double *sc = scratch + v * 64 + r;
            double tot = (sc[0] + sc[16]) + (sc[32] + sc[48]);
            tot = dpp_add<0x111, 0xf>(tot); // row_shr:1
            tot = dpp_add<0x112, 0xf>(tot); // row_shr:2
            tot = dpp_add<0x114, 0xf>(tot); // row_shr:4
            tot = dpp_add<0x118, 0xf>(tot); // row_shr:8 -> lane 15 of row v holds value v
            const int64_t l = lbase + loc;
            if (r == 15 && v < NV && l < l_end) P.spec[l * NV + v] = tot;
            __builtin_amdgcn_wave_barrier();
        }
    };

    // ---- main loop ------------------------------------------------------------------------------
    // Rolling software pipeline over the (k-step, tile-slot) sequence of a stage: the fragment pair
    // of item q + D is requested before the MFMA of item q is issued (ring of D + 1 pairs in
    // registers).  The stage barrier sits D items before the end of the stage, so every wave still
    // holds D MFMAs' worth of operands when it arrives: the barrier's arrival skew and the first LDS
    // round trip of the next stage hide behind them.
    // W >= 0 (SPEC13 kernels: one block of 13 tiles, 193..208 pools, LDS pitch 208): the body is
    // instantiated once per wave with COMPILE-TIME tile coordinates, so every fragment address is
    // "lane base + immediate" (no per-item address arithmetic, A/B pairs merge into ds_read2_b64).
    // W = -1: run-time tile tables, any shape.
    // TPW = tile slots per wave actually run (2..6): a work-group with few tiles (small n, or a
    // diagonal block pair) would otherwise spend most of its matrix-core time on dead slots.
    auto run = [&](auto wc, auto tc) {
        constexpr int W = decltype(wc)::value;
        constexpr int TPW = decltype(tc)::value;
        constexpr int KS = KIN_KC / 4;
        constexpr int NQ = KS * TPW; // items per stage
        constexpr int D = KIN_RING_D, R = KIN_RING_D + 1;
        static_assert(NQ % R == 0 && NQ > 2 * D, "ring indexing assumes the stage length is a multiple of the ring");
        double4_t acc[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u] = (double4_t){0.0, 0.0, 0.0, 0.0};
#ifndef KIN_SHARED_FRAGS
#define KIN_SHARED_FRAGS 1
#endif
        if constexpr (W >= 0 && KIN_SHARED_FRAGS) {
            // ---- 13-tile shape with shared fragments: per k-step the wave reads its <= 6 distinct fragments once
            // (two register sets: the set of step s + 1 is requested before the MFMAs of step s are issued) and
            // feeds its <= 6 tiles from them.  The stage barrier sits in front of the LAST k-step's MFMAs: their
            // operands are already in registers, and the first reads of the next stage hide behind them.
            constexpr int NC = kin13_ncols(W), NT = kin13_ntiles(W);
            static_assert(NC <= KIN13_MAXC && KS % 2 == 0, "fragment sets alternate with the k-step parity");
            double fr[2][KIN13_MAXC];
            const double *lb = lds + kq * ldsld + fi;
            // which of this wave's tiles are strip tiles (r, 12), r < 12
            constexpr bool has_strip = [] { bool h = false; for (int u = 0; u < 6; ++u) h = h || (KIN13_T[W][u][0] >= 0 && KIN13_T[W][u][0] < 12 && KIN13_T[W][u][1] == 12); return h; }();
            double bs[2][2];                          // strip B operands: pools 192 + 4 c + (lane & 3), c = 0, 1
            const double *lbs = lds + kq * ldsld + 192 + (lane & 3);
            // diagonal tiles: the B operands of the block diagonals d = 1, 2 = the tile's fragment rotated by 4 d lanes inside its
            // 16-lane row.  Read from LDS at the rotated address (a DPP rotation is VALU work on the port the MFMAs issue from:
            // measured, it ate the saving)
            constexpr int ND = (SMALL & 1) ? kin13_ndiag(W) : 0;
            double bd[2][ND > 0 ? ND : 1][2];
            const double *lbr1 = lds + kq * ldsld + ((fi + 4) & 15), *lbr2 = lds + kq * ldsld + ((fi + 8) & 15);
            auto load_set = [&](const double *bufbase, auto sc, auto pc) __attribute__((always_inline)) {
                constexpr int s2 = decltype(sc)::value, par = decltype(pc)::value;
                static_for<NC>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int ci = decltype(cc)::value;
                    fr[par][ci] = bufbase[4 * s2 * 208 + 16 * kin13_col(W, ci)];
                });
                if constexpr ((SMALL & 2) != 0 && has_strip) {
                    const double *bb = lbs + (bufbase - lb);
                    bs[par][0] = bb[4 * s2 * 208];
                    bs[par][1] = bb[4 * s2 * 208 + 4];
                }
                if constexpr (ND > 0) {
                    static_for<NT>([&](auto uc) __attribute__((always_inline)) {
                        constexpr int u = decltype(uc)::value;
                        if constexpr (KIN13_T[W][u][0] == KIN13_T[W][u][1]) {
                            constexpr int di = kin13_diag_index(W, u);
                            bd[par][di][0] = (lbr1 + (bufbase - lb))[4 * s2 * 208 + 16 * KIN13_T[W][u][0]];
                            bd[par][di][1] = (lbr2 + (bufbase - lb))[4 * s2 * 208 + 16 * KIN13_T[W][u][0]];
                        }
                    });
                }
            };
            auto mfma_set = [&](auto pc) __attribute__((always_inline)) {
                constexpr int par = decltype(pc)::value;
                static_for<NT>([&](auto uc) __attribute__((always_inline)) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int ti = KIN13_T[W][u][0], tj = KIN13_T[W][u][1];
                    const double fa = fr[par][kin13_slot(W, ti)];
                    if constexpr ((SMALL & 1) != 0 && ti == tj) {
                        // diagonal tile: block diagonals d = 0, 1, 2 of its 4 x 4 grid of 4 x 4 blocks
                        acc[u][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fa, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, bd[par][kin13_diag_index(W, u)][0], acc[u][1], 0, 0, 0);
                        acc[u][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, bd[par][kin13_diag_index(W, u)][1], acc[u][2], 0, 0, 0);
                    } else if constexpr ((SMALL & 2) != 0 && tj == 12 && ti != tj) {
                        acc[u][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, bs[par][0], acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, bs[par][1], acc[u][1], 0, 0, 0);
                    } else {
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fr[par][kin13_slot(W, tj)], acc[u], 0, 0, 0);
                    }
                });
            };
            if (nstages > 0) load_set(lb, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            constexpr int spec_step = (W / 4) % KS; // FUSE: the four waves of a SIMD do their per-locus sums in different k-steps
            for (int c = 0; c < nstages; ++c) {
                const bool more = (c + 1) < nstages;
                if (more) stage_load(c + 1);
                const double *buf = lds + (c & 1) * bufsz;
                const double *b0 = lb + (c & 1) * bufsz;
                const double *b1 = lb + ((c + 1) & 1) * bufsz;
                static_for<KS>([&](auto sc) __attribute__((always_inline)) {
                    constexpr int s2 = decltype(sc)::value;
                    if constexpr (FUSE && s2 == spec_step) spec_pass(buf, l_begin + (int64_t)c * KIN_KC);
                    if constexpr (s2 + 1 < KS) {
                        load_set(b0, std::integral_constant<int, s2 + 1>{}, std::integral_constant<int, (s2 + 1) & 1>{});
                    } else {
                        if (more) stage_store((c + 1) & 1);
                        __syncthreads();
                        if (more) load_set(b1, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                    }
                    mfma_set(std::integral_constant<int, s2 & 1>{});
                });
            }
            double *slab = P.slabs + (size_t)bx * P.npad * P.npad;
            static_for<NT>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                constexpr int ti = KIN13_T[W][u][0], tj = KIN13_T[W][u][1];
                constexpr int r0 = 16 * ti, c0 = 16 * tj;
                // 4 x 4 blocks: this lane holds D[blk][i][j] with i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3
                const int bi = lane >> 4, bb = (lane >> 2) & 3, bj = lane & 3;
                if constexpr ((SMALL & 1) != 0 && ti == tj) {
                    // d = 0: block (bb, bb); d = 1: (bb, bb + 1), the wrapped (3, 0) stored as its transpose (0, 3); d = 2: (0, 2), (1, 3)
                    slab[(size_t)(r0 + 4 * bb + bi) * P.npad + c0 + 4 * bb + bj] = acc[u][0];
                    if (bb < 3) slab[(size_t)(r0 + 4 * bb + bi) * P.npad + c0 + 4 * (bb + 1) + bj] = acc[u][1];
                    else slab[(size_t)(r0 + bj) * P.npad + c0 + 12 + bi] = acc[u][1];
                    if (bb < 2) slab[(size_t)(r0 + 4 * bb + bi) * P.npad + c0 + 4 * (bb + 2) + bj] = acc[u][2];
                } else if constexpr ((SMALL & 2) != 0 && tj == 12 && ti != tj) {
                    slab[(size_t)(r0 + 4 * bb + bi) * P.npad + c0 + bj] = acc[u][0];
                    slab[(size_t)(r0 + 4 * bb + bi) * P.npad + c0 + 4 + bj] = acc[u][1];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(size_t)(r0 + kq + 4 * r) * P.npad + c0 + fi] = acc[u][r];
                }
            });
            return;
        }
        double fa[R], fb[R];
        const double *lanebase = lds + kq * ldsld + fi; // + buffer + 4 s ldsld + tile column
        auto frag_load = [&](const double *bufbase, auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int s = q / TPW, u = q % TPW;
            if constexpr (W >= 0) {
                if constexpr (W + KIN_WAVES * u >= 91) return; // no such tile: the slot is simply skipped
                constexpr int t = kin_slot_tile(W, u, 13);
                constexpr int ao = 4 * s * 208 + 16 * kin_tri_ti(t, 13);
                constexpr int bo = 4 * s * 208 + 16 * kin_tri_tj(t, 13);
                fa[q % R] = bufbase[ao];
#ifdef KIN_EXP_HALF
                fb[q % R] = fa[q % R]; // experiment: half the LDS reads (wrong numbers, timing only)
                (void)bo;
#elif defined(KIN_EXP_QUARTER)
                if constexpr (q % 2 == 0) fa[q % R] = bufbase[ao]; else fa[q % R] = fa[(q + R - 1) % R];
                fb[q % R] = fa[q % R];
                (void)bo;
#else
                fb[q % R] = bufbase[bo];
#endif
            } else {
                if constexpr (SKIP) { if (!live[u]) return; } // (wave-uniform) a slot beyond this wave's tiles: neither its reads nor its matrix cycles
                const double *row = bufbase + 4 * s * ldsld;
                fa[q % R] = row[acol[u]];
                fb[q % R] = row[bcol[u]];
            }
        };
        auto mfma_item = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int u = q % TPW;
            if constexpr (W >= 0 && W + KIN_WAVES * u >= 91) return;
            if constexpr (W < 0 && SKIP) { if (!live[u]) return; }
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[q % R], fb[q % R], acc[u], 0, 0, 0);
        };
        if (nstages > 0)
            static_for<D>([&](auto qc) { frag_load(lanebase, qc); });
        for (int c = 0; c < nstages; ++c) {
            const bool more = (c + 1) < nstages;
            if (more) stage_load(c + 1);
            const double *buf = lds + (c & 1) * bufsz;
            const double *fb0 = lanebase + (c & 1) * bufsz;
            const double *fb1 = lanebase + ((c + 1) & 1) * bufsz;
            // FUSE: the per-locus sums are VALU/LDS work that only overlaps the matrix cores if the waves of a
            // SIMD do it at DIFFERENT times: with compile-time wave ids (W >= 0) wave w of SIMD w % 4 runs its
            // spec_pass after item (w / 4) * NQ / 4 of the MFMA block instead of all four at its end.
#ifndef KIN_SPEC_STAGGER
#define KIN_SPEC_STAGGER 1
#endif
            constexpr int spec_at = (FUSE && W >= 0 && KIN_SPEC_STAGGER) ? ((W / 4) % 4) * ((NQ - D) / 4) : -1;
            static_for<NQ - D>([&](auto qc) {
                if constexpr (decltype(qc)::value == spec_at) spec_pass(buf, l_begin + (int64_t)c * KIN_KC);
                frag_load(fb0, std::integral_constant<int, decltype(qc)::value + D>{});
                mfma_item(qc);
            });
            if (more) stage_store((c + 1) & 1);
            if (FUSE && spec_at < 0) spec_pass(buf, l_begin + (int64_t)c * KIN_KC); // staging registers are free again here
            __syncthreads();
            static_for<D>([&](auto dc) {
                constexpr int q = NQ - D + decltype(dc)::value;
                if (more) frag_load(fb1, std::integral_constant<int, q + D - NQ>{});
                mfma_item(std::integral_constant<int, q>{});
            });
        }
        // ---- write this workgroup's partial tiles -----------------------------------------------
        // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.
        double *slab = P.slabs + (size_t)bx * P.npad * P.npad;
        static_for<TPW>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            bool on;
            int r0, c0;
            if constexpr (W >= 0) {
                constexpr int t = W + KIN_WAVES * u;
                on = t < 91;
                r0 = 16 * kin_tri_ti(kin_slot_tile(W, u, 13), 13);
                c0 = 16 * kin_tri_tj(kin_slot_tile(W, u, 13), 13);
            } else {
                on = live[u]; r0 = orow[u]; c0 = ocol[u];
            }
            if (on) {
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(size_t)(r0 + kq + 4 * r) * P.npad + c0 + fi] = acc[u][r];
            }
        });
    };
    if constexpr (SPEC13) {
        switch (wave) {
        case 0: run(std::integral_constant<int, 0>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 1: run(std::integral_constant<int, 1>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 2: run(std::integral_constant<int, 2>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 3: run(std::integral_constant<int, 3>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 4: run(std::integral_constant<int, 4>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 5: run(std::integral_constant<int, 5>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 6: run(std::integral_constant<int, 6>{}, std::integral_constant<int, KIN_TPW>{}); break;
#if KIN_WAVES_DEF == 8
        default: run(std::integral_constant<int, 7>{}, std::integral_constant<int, KIN_TPW>{}); break;
#else
        case 7: run(std::integral_constant<int, 7>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 8: run(std::integral_constant<int, 8>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 9: run(std::integral_constant<int, 9>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 10: run(std::integral_constant<int, 10>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 11: run(std::integral_constant<int, 11>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 12: run(std::integral_constant<int, 12>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 13: run(std::integral_constant<int, 13>{}, std::integral_constant<int, KIN_TPW>{}); break;
        case 14: run(std::integral_constant<int, 14>{}, std::integral_constant<int, KIN_TPW>{}); break;
        default: run(std::integral_constant<int, 15>{}, std::integral_constant<int, KIN_TPW>{}); break;
#endif
        }
    } else {
        constexpr std::integral_constant<int, -1> any{};
        switch ((ntiles + KIN_WAVES - 1) / KIN_WAVES) {
        case 0: case 1: case 2: run(any, std::integral_constant<int, 2>{}); break;
        case 3: run(any, std::integral_constant<int, 3>{}); break;
        case 4: run(any, std::integral_constant<int, 4>{}); break;
        case 5: run(any, std::integral_constant<int, 5>{}); break;
        default: run(any, std::integral_constant<int, KIN_TPW>{}); break;
        }
    }
}

// Sum the per-workgroup slabs in a FIXED order and mirror: S[i][j] = S[j][i] = sum_w slab_w[i][j] for
// i <= j (only upper-triangular tiles were written).  256 threads = 64 columns x 4 slab groups: group g
// adds slabs g, g+4, ... in order, then (g0 + g1) + (g2 + g3) -- deterministic, no atomics, and four
// times the loads in flight of a one-thread-per-entry loop (the slabs are 88 MB at n = 200).
constexpr double KIN_PAIR_FLOOR = 2.7; // slots' worth of time a block pair costs at least (stage hand-over)
struct KinCounts { int blk, nb, n[6]; }; // weighted pairs: slabs that hold the tiles of block pair q (row-major upper triangle); blk = 0: all
__global__ __launch_bounds__(256) void k_kinship_reduce(const double *__restrict__ slabs, int nslabs, int npad, int n,
                                                        double add_const, double *__restrict__ S, const KinCounts C) {
    __shared__ double part[4][64];
    const int jj = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jj;
    const int i = blockIdx.y;
    const bool on = i < n && j < n && i <= j;
    double s = 0.0;
    if (on && C.blk) {
        const int bi = i / C.blk, bj = j / C.blk;
        nslabs = C.n[bi * C.nb - bi * (bi - 1) / 2 + (bj - bi)];
    }
    if (on) {
        const size_t off = (size_t)i * npad + j;
        const size_t stride = (size_t)npad * npad;
        for (int w = grp; w < nslabs; w += 4) s += slabs[w * stride + off];
    }
    part[grp][jj] = s;
    __syncthreads();
    if (on && grp == 0) {
        const double t = ((part[0][jj] + part[1][jj]) + (part[2][jj] + part[3][jj])) + add_const;
        S[(size_t)i * n + j] = t;
        S[(size_t)j * n + i] = t;
    }
}

} // namespace

#ifndef KIN_LAUNCH_NAME
#define KIN_LAUNCH_NAME pg_launch_kinship
#endif
int KIN_LAUNCH_NAME(pg_ctx *ctx, const double *G, int64_t p, int n, int64_t ld, double *S,
                      bool add_intercept, int kid, bool allow_fuse) {
#if KIN_WAVES_DEF == 16
    // Up to 64 pools (<= 10 tiles) a 16-wave workgroup leaves most of its waves without a tile: the same kernel built for 8-wave
    // workgroups (pg_kinship_w8.hip) takes over -- 1.36 -> 1.01 ms per 4 M loci at 48 and 64 pools; from 100 pools on it is no
    // faster, at 150 slower (profiles/r03_kinship_small_n.log).  POOLGEN_KIN_NO_W8=1: A/B runs.
    if (n <= 64 && !std::getenv("POOLGEN_KIN_NO_W8")) return pg_launch_kinship_w8(ctx, G, p, n, ld, S, add_intercept, kid, allow_fuse);
#endif

    PG_CHECK(ctx, G && S, "kinship: null pointer");
    PG_CHECK(ctx, p > 0 && n > 0, "kinship: need p > 0 and n > 0 (p=%lld n=%d)", (long long)p, n);
    PG_CHECK(ctx, ld >= n && (ld % 2) == 0, "kinship: ld (%lld) must be even and >= n (%d)",
             (long long)ld, n);
    PG_CHECK(ctx, (reinterpret_cast<uintptr_t>(G) & 15) == 0, "kinship: G must be 16-byte aligned");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int cus = ctx->cus;

    KinParams P;
    P.G = G; P.p = p; P.ld = ld; P.n = n;
    P.T = (n + 15) / 16;
    P.npad = P.T * 16;
    if (P.T <= 13) { P.Tb = P.T; P.nb = 1; }
    else { P.Tb = 8; P.nb = (P.T + 7) / 8; }
    // nb >= 4: 64 + 2 * ceil(36 / (nb - 1)) <= 96 slots; nb == 2: measured -- a win at T = 15, 16 (7.6 -> 6.9 ms at n = 240..256,
    // 4 M loci), a loss at T = 14 where the diagonal workgroups stage few columns
    P.merged = ((P.nb >= 4 || (P.nb == 2 && P.T >= 15)) && !std::getenv("POOLGEN_KIN_NO_MERGE")) ? 1 : 0;
    if (!P.merged && P.nb >= 2 && P.nb <= 3 && !std::getenv("POOLGEN_KIN_NO_WEIGHTS")) {
        // the block size that runs the fewest tile slots over all pairs (T = 17: blocks of 6, 6, 5 tile columns = 13 slot units
        // where 8, 8, 1 ran 16)
        // (a pair with two slots is bound by its stage hand-over, not by its MFMAs: it costs about 2.7 slots' worth -- measured
        //  at T = 14, where blocks of 7 + 7 (2, 2 and 4 slots) ran 5.98 ms against 5.46 ms for 8 + 6 (3, 2 and 3))
        double best = 1e30;
        int best_tb = P.Tb;
        for (int tb = (P.T + P.nb - 1) / P.nb; tb <= 8; ++tb) {
            double units = 0.0;
            for (int bi = 0; bi < P.nb; ++bi)
                for (int bj = bi; bj < P.nb; ++bj) {
                    const int Ta = std::min(tb, P.T - bi * tb), Tbb = std::min(tb, P.T - bj * tb);
                    if (Ta <= 0 || Tbb <= 0) { units += 1e6; continue; }
                    const int tiles = bi == bj ? Ta * (Ta + 1) / 2 : Ta * Tbb;
                    units += std::max(KIN_PAIR_FLOOR, (double)((tiles + KIN_WAVES - 1) / KIN_WAVES));
                }
            if (units <= best) { best = units; best_tb = tb; }
        }
        P.Tb = best_tb;
    }
    P.split = (P.merged && P.nb == 2) ? 2 : 1;                                               // nb == 2: all 136 tiles, 68 + 68
    const int npairs = (P.merged ? P.nb * (P.nb - 1) / 2 : P.nb * (P.nb + 1) / 2) * P.split;
    int nslab = cus / npairs;
    if (const char *e = std::getenv("POOLGEN_KIN_SLAB_MULT")) nslab *= std::max(1, std::atoi(e)); // experiments: workgroups per CU
    if (nslab < 1) nslab = 1;
    const int64_t max_slabs = (p + KIN_KC - 1) / KIN_KC;
    if (nslab > max_slabs) nslab = (int)max_slabs;
    P.loci_per_wg = (p + nslab - 1) / nslab;
    P.loci_per_wg = (P.loci_per_wg + KIN_KC - 1) / KIN_KC * KIN_KC;
    // a slab must be addressable with the 32-bit offsets of its buffer descriptor
    const int64_t max_loci_per_wg = ((int64_t)0xFFFFFFF0 / (ld * 8)) / KIN_KC * KIN_KC - KIN_KC;
    if (P.loci_per_wg > max_loci_per_wg) P.loci_per_wg = max_loci_per_wg;
    nslab = (int)((p + P.loci_per_wg - 1) / P.loci_per_wg);

    P.npairs = npairs; P.nslab = nslab;
    P.skip_dead = (!std::getenv("POOLGEN_KIN_NO_SKIP") && P.nb <= 2) ? 1 : 0; // (from 4 blocks on the lists are full: 500 pools 10.17 -> 10.36 ms with it)
    P.xcd_spx = 0;
    if (npairs > 1 && cus % 8 == 0 && !std::getenv("POOLGEN_KIN_NO_XCD")) {
        const int spx = (cus / 8) / npairs;        // whole slabs (all their pairs) that fit one XCD's CUs
        if (spx >= 1 && 8 * spx <= nslab) P.xcd_spx = spx;
    }
    for (int q = 0; q < 6; ++q) { P.wq_n[q] = 0; P.wq_first[q] = 0; P.wq_len[q] = 0; }
    KinCounts KC;
    KC.blk = 0; KC.nb = P.nb;
    for (int q = 0; q < 6; ++q) KC.n[q] = 0;
    int grid1d = 0;
    if (P.nb >= 2 && P.nb <= 3 && !P.merged && !std::getenv("POOLGEN_KIN_NO_WEIGHTS")) {
        // tile slots a wave of pair (bi, bj) runs per k-step (the kernel's TPW) + a constant for its share of the staging
        double wgt[6], wsum = 0.0;
        int q = 0;
        for (int bi = 0; bi < P.nb; ++bi)
            for (int bj = bi; bj < P.nb; ++bj, ++q) {
                const int Ta = std::min(P.Tb, P.T - bi * P.Tb), Tbb = std::min(P.Tb, P.T - bj * P.Tb);
                const int tiles = bi == bj ? Ta * (Ta + 1) / 2 : Ta * Tbb;
                wgt[q] = std::max(KIN_PAIR_FLOOR, (double)((tiles + KIN_WAVES - 1) / KIN_WAVES)) + 0.35;
                wsum += wgt[q];
            }
        int used = 0;
        for (q = 0; q < npairs; ++q) { P.wq_n[q] = std::max(1, (int)(cus * wgt[q] / wsum)); used += P.wq_n[q]; }
        for (int left = cus - used; left > 0; --left) { // the CUs left over go where the slabs are longest
            int best = 0;
            for (q = 1; q < npairs; ++q)
                if (wgt[q] / P.wq_n[q] > wgt[best] / P.wq_n[best]) best = q;
            ++P.wq_n[best];
        }
        int first = 0, maxn = 0;
        for (q = 0; q < npairs; ++q) {
            int64_t len = (p + P.wq_n[q] - 1) / P.wq_n[q];
            len = (len + KIN_KC - 1) / KIN_KC * KIN_KC;
            if (len > max_loci_per_wg) len = max_loci_per_wg;
            P.wq_len[q] = len;
            P.wq_n[q] = (int)((p + len - 1) / len);
            P.wq_first[q] = first;
            first += P.wq_n[q];
            maxn = std::max(maxn, P.wq_n[q]);
            KC.n[q] = P.wq_n[q];
        }
        grid1d = first;
        nslab = maxn;
        P.nslab = nslab;
        P.xcd_spx = 0;
        P.skip_dead = 0;
        KC.blk = P.Tb * 16;
    }
    const size_t slab_bytes = (size_t)nslab * P.npad * P.npad * sizeof(double);
    int rc = pg_ws_reserve(ctx, slab_bytes);
    if (rc) return rc;
    P.slabs = static_cast<double *>(ctx->ws);

    // worst-case LDS row: two blocks of Tb tiles (+16 pad)
    int ldsld = (P.nb == 1) ? P.Tb * 16 : 2 * P.Tb * 16;
    if ((ldsld & 31) != 16) ldsld += 16;
    // fused speculative intercept-only sums: only in the single-block path and with phenotypes announced
    const size_t fuse_lds = (size_t)2 * KIN_KC * ldsld * 8 + ((size_t)KIN_FUSE_MAXK * 256 + (size_t)KIN_WAVES * 4 * 64) * 8;
    const bool fuse = allow_fuse && fuse_lds <= 160 * 1024 && P.nb == 1 && ctx->ph_n == n && ctx->ph_k >= 1 && ctx->ph_k <= KIN_FUSE_MAXK &&
                      ctx->ph_ytil_dev != nullptr;
    P.ytil = nullptr; P.spec = nullptr; P.k = 0;
    ctx->spec_valid = false;
    if (fuse) {
        const size_t need = (size_t)p * (2 + ctx->ph_k) * sizeof(double);
        if (need > ctx->spec_cap) {
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->spec_dev) PG_HIP(ctx, hipFree(ctx->spec_dev));
            ctx->spec_dev = nullptr; ctx->spec_cap = 0;
            PG_HIP(ctx, hipMalloc((void **)&ctx->spec_dev, need));
            ctx->spec_cap = need;
        }
        P.ytil = ctx->ph_ytil_dev; P.spec = ctx->spec_dev; P.k = ctx->ph_k;
    }
    const size_t shmem = (size_t)2 * KIN_KC * ldsld * sizeof(double) + (fuse ? ((size_t)KIN_FUSE_MAXK * 256 + (size_t)KIN_WAVES * 4 * 64) * sizeof(double) : 0);
    const int max_cols = (P.nb == 1) ? P.Tb * 16 : 2 * P.Tb * 16; // staged pools per locus row
    PG_CHECK(ctx, KIN_KC * (max_cols / 2) <= KIN_PPT * KIN_THREADS, "kinship: staging overflow");
    const bool spec13 = (P.nb == 1 && P.T == 13);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        if (grid1d) hipLaunchKernelGGL(kern, dim3(grid1d), dim3(KIN_THREADS), shmem, ctx->stream, P);
        else if (P.xcd_spx) hipLaunchKernelGGL(kern, dim3(8 * ((P.xcd_spx * npairs) + ((nslab - 8 * P.xcd_spx) * npairs + 7) / 8)), dim3(KIN_THREADS), shmem,
                                          ctx->stream, P);
        else hipLaunchKernelGGL(kern, dim3(nslab, npairs), dim3(KIN_THREADS), shmem, ctx->stream, P);
        return hipSuccess;
    };
    pg_prof_begin(ctx, kid);
    hipError_t le;
    // 4 x 4 x 4 blocks for the narrow tiles of the 13-tile shape (see k_kinship_syrk): the diagonal tiles (bit 1) and the last
    // column when it holds <= 8 pools (bit 2).  Measured at 200 pools x 10 M loci (tools/bench_kin_ab.py, one box, ms):
    //   bits          0      1      2      3
    //   fused sums   7.62   7.59   7.31   7.23     (bit 1 costs the fused kernel 19 spilled registers and still pays)
    //   plain        6.84   6.80   6.41   6.30
    // POOLGEN_KIN_SMALL=<bits> overrides (A/B timing).
    int sm = !spec13 ? 0 : ((n <= 200 ? 2 : 0) | 1);
    if (const char *e = std::getenv("POOLGEN_KIN_SMALL")) sm = spec13 ? (std::atoi(e) & (n <= 200 ? 3 : 1)) : 0;
    if (std::getenv("POOLGEN_KIN_NO_SMALL")) sm = 0;
    auto pick = [&](auto fz) -> hipError_t {
        constexpr bool FZ = decltype(fz)::value;
        if (!spec13) return P.skip_dead ? launch(k_kinship_syrk<FZ, false, 0, true>) : launch(k_kinship_syrk<FZ, false, 0, false>);
        switch (sm) {
        case 1: return launch(k_kinship_syrk<FZ, true, 1>);
        case 2: return launch(k_kinship_syrk<FZ, true, 2>);
        case 3: return launch(k_kinship_syrk<FZ, true, 3>);
        default: return launch(k_kinship_syrk<FZ, true, 0>);
        }
    };
    le = fuse ? pick(std::true_type{}) : pick(std::false_type{});
    PG_HIP(ctx, le);
    pg_prof_end(ctx);
    if (fuse) {
        ctx->spec_G = G; ctx->spec_p = p; ctx->spec_ld = ld; ctx->spec_n = n; ctx->spec_k = ctx->ph_k;
        ctx->spec_valid = true;
    }
    PG_HIP(ctx, hipGetLastError());
    pg_prof_begin(ctx, PG_K_KINSHIP_REDUCE);
    hipLaunchKernelGGL(k_kinship_reduce, dim3((n + 63) / 64, n), dim3(256), 0, ctx->stream,
                       P.slabs, nslab, P.npad, n, add_intercept ? 1.0 : 0.0, S, KC);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());
    return PG_OK;
}

#ifndef KIN_NO_EXPORTS // (pg_kinship_w8.hip includes this file for its launcher only)
extern "C" int pg_kinship_partial_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n,
                                      int64_t ld, double *S_dev) {
    if (!ctx) return PG_ERR_INVALID;
    return pg_launch_kinship(ctx, G_dev, p, n, ld, S_dev, false, PG_K_KINSHIP, true);
}

extern "C" int pg_gp_xxt_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld,
                             double *XXt_dev) {
    if (!ctx) return PG_ERR_INVALID;
    // X = [1 | G^T]  =>  X X^T = 1 1^T + sum_l g_l g_l^T
    return pg_launch_kinship(ctx, G_dev, p, n, ld, XXt_dev, true, PG_K_GP_XXT);
}

extern "C" int pg_set_phenotypes(pg_ctx *ctx, int n, const double *Y, int k) {
    if (!ctx) return PG_ERR_INVALID;
    ctx->spec_valid = false;
    if (n > 0 && Y && ctx->ph_n == n && ctx->ph_k == k && ctx->ph_ytil_dev && ctx->ph_Y.size() == (size_t)n * k &&
        std::memcmp(ctx->ph_Y.data(), Y, sizeof(double) * n * k) == 0)
        return PG_OK; // same phenotypes as last time: the centred copy is already on the device
    ctx->ph_n = 0; ctx->ph_k = 0; ctx->ph_Y.clear();
    if (n == 0 || !Y) return PG_OK; // switched off
    PG_CHECK(ctx, n >= 2 && k >= 1, "set_phenotypes: bad shape n=%d k=%d", n, k);
    if (k > 4 || n > 256) return PG_OK; // not an error: the regular two-pass path is used
    for (int i = 0; i < n * k; ++i)
        PG_CHECK(ctx, Y[i] == Y[i], "set_phenotypes: phenotype matrix contains NaN; remove pools with "
                                    "missing phenotypes first (gwas/ols.rs:287)");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> yt((size_t)k * 256, 0.0);
    for (int t = 0; t < k; ++t) {
        double mu = 0.0;
        for (int i = 0; i < n; ++i) mu += Y[(size_t)i * k + t];
        mu /= n;
        double syy = 0.0;
        for (int i = 0; i < n; ++i) {
            const double y = Y[(size_t)i * k + t] - mu;
            yt[(size_t)t * 256 + i] = y;
            syy += y * y;
        }
        ctx->ph_syy[t] = syy;
    }
    if (!ctx->ph_ytil_dev) PG_HIP(ctx, hipMalloc((void **)&ctx->ph_ytil_dev, sizeof(double) * 4 * 256));
    PG_HIP(ctx, hipMemcpyAsync(ctx->ph_ytil_dev, yt.data(), sizeof(double) * k * 256, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ph_Y.assign(Y, Y + (size_t)n * k);
    ctx->ph_n = n; ctx->ph_k = k;
    return PG_OK;
}
#endif // KIN_NO_EXPORTS
