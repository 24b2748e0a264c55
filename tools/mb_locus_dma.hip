// mb_locus_dma.hip -- prototype of the count operators' streaming loop with LDS-DMA staging (round 3).
//
// What it prices before k_locus_first is rewritten: counts[L][n][6] u32 rows (24 n bytes) streamed by one lane per locus,
// HBM -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write), a wave-private ring of NSLOT slots of
// SB bytes per row, pools read back with ds_read_b64 at compile-time offsets (a ring turn = NSLOT * SB bytes = a whole
// number of 24-byte pools, so every LDS offset of the unrolled turn is an immediate), with the filter + operator
// arithmetic of a pool in place (WORK 1) or an integer checksum only (WORK 0, validated on the host).
//
// Rows are taken in UNITS of 64 rows of one alignment class (rows r0, r0 + period, ...), as k_locus_first does: every
// lane sees the same pool boundaries, every 128-byte line is requested by one wave only.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mb_locus_dma.hip -o tools/mb_locus_dma
// Run:   tools/mb_locus_dma [pools] [loci]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ double recip_for_div(double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double r1 = fma(fma(-b, r0, 1.0), r0, r0);
    return fma(fma(-b, r1, 1.0), r1, r1);
}
__device__ __forceinline__ double div_by(double a, double b, double r) {
    const double q0 = a * r;
    return fma(fma(-b, q0, a), r, q0);
}

struct UnitD { // everything wave-uniform
    uint32_t base_lo, base_hi; // SB-aligned address of the first piece of row r0
    int delta;                 // offset of row r0's first byte inside that piece
    int nh;                    // pieces per row that hold bytes of the row
    int rows;                  // valid rows (<= 64)
    int64_t r0;
};

// NT 1: `nt` on the DMA requests.  STG 1: no LDS-DMA -- raw buffer loads into 4 * IPS registers, written to the ring slot one step
// later (one slot in flight per wave, in registers: the shape of round 2's k_locus_first), for an A/B of the two request paths.
// FL bit 0: do not request pieces past the row's end (STG only; the DMA form needs a fixed number of requests per step)
//    bit 1: write the operators' outputs (144 bytes per locus: n_out, 5 ids, 5 mean frequencies, 5 statistics, 5 p-values)
//    bit 2: a wave takes the `period` classes of a group one after the other (the line two neighbouring rows share is then
//           requested twice within microseconds by the same CU) instead of the block's waves taking them side by side
template <int SB, int NSLOT, int WORK, int WPB, int MINW, int NT = 0, int STG = 0, int FL = 0>
__global__ __launch_bounds__(64 * WPB, MINW) void k_stream(const uint32_t *__restrict__ counts, const double *__restrict__ wy,
                                                          int64_t L, int n, int period, double *__restrict__ out) {
    constexpr int TB = SB * NSLOT;       // bytes of a row per ring turn
    static_assert(TB % 24 == 0 || WORK == 0, "a ring turn must hold a whole number of pools");
    constexpr int PPT = TB / 24;         // pools per turn
    constexpr int HPS = SB / 64;         // 64-byte halves per slot
    constexpr int IPS = 4 * HPS;         // DMA instructions per slot: 16 rows x 64 bytes each
    constexpr int RPI = 16;              // rows per DMA instruction
    constexpr int SLOTB = 64 * SB;       // LDS bytes of a slot
    constexpr int NJ = 5;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PSLOT = STG ? 1 : NSLOT; // physical slots per wave: with register staging every logical slot is the same memory
    char *ring = lds + wave * (PSLOT * SLOTB);
    const double *tab = reinterpret_cast<const double *>(lds + WPB * PSLOT * SLOTB); // [n][2]: w_i, y_i
    {
        double *t = reinterpret_cast<double *>(lds + WPB * PSLOT * SLOTB);
        for (int i = threadIdx.x; i < 2 * n; i += 64 * WPB) t[i] = wy[i];
        __syncthreads();
    }
    // LDS byte address of this wave's ring (wave-uniform): what M0 wants
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ring);
    const uint32_t rowb = (uint32_t)n * 24u;
    const int pshift = __builtin_ctz(period);
    const uint32_t pstride = (uint32_t)period * rowb;
    const int64_t ngroups = (L + 64 * (int64_t)period - 1) >> (6 + pshift);
    const int64_t nunits = ngroups * period;
    const int cu = (FL & 4) ? period : (period > WPB ? period : WPB);
    const int64_t nchunks = (nunits + cu - 1) / cu;
    const int64_t chunk0 = (FL & 4) ? (int64_t)blockIdx.x * WPB + wave : (int64_t)blockIdx.x;
    const int64_t chunk_step = (FL & 4) ? (int64_t)gridDim.x * WPB : (int64_t)gridDim.x;
    const int q0 = (FL & 4) ? 0 : wave, q_step = (FL & 4) ? 1 : WPB;

    // DMA role of this lane: row rr of the instruction's RPI rows, 16-byte piece pp of the slot
    const int rr = lane % RPI, pp = lane / RPI;
    // this lane's own row (= lane) inside a slot
    const char *cell = ring + (lane / RPI) * (HPS * 1024) + (lane % RPI) * 16;

    auto make_unit = [&](int64_t u) {
        UnitD d;
        const int64_t g = u >> pshift;
        d.r0 = (g << (6 + pshift)) + (u - (g << pshift));
        const uint64_t addr0 = reinterpret_cast<uint64_t>(counts) + (uint64_t)d.r0 * rowb;
        d.delta = __builtin_amdgcn_readfirstlane((int)((uint32_t)addr0 & (uint32_t)(SB - 1)));
        d.nh = (int)(((uint32_t)d.delta + rowb + (uint32_t)(SB - 1)) / (uint32_t)SB);
        int64_t rows = d.r0 < L ? (L - d.r0 + period - 1) >> pshift : 0;
        d.rows = __builtin_amdgcn_readfirstlane((int)(rows < 64 ? rows : 64));
        const uint64_t basev = addr0 & ~(uint64_t)(SB - 1);
        d.base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)basev);
        d.base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(basev >> 32));
        return d;
    };

    struct Cursor { int64_t chunk; int q; bool valid; UnitD D; };
    auto unit_of = [&](int64_t ch, int q_) { return ch * cu + q_; };
    auto next_unit = [&](Cursor &c) {
        c.q += q_step;
        if (c.q >= cu || unit_of(c.chunk, c.q) >= nunits) { c.q = q0; c.chunk += chunk_step; }
        const int64_t u = unit_of(c.chunk, c.q);
        c.valid = c.chunk < nchunks && u < nunits;
        if (c.valid) c.D = make_unit(u);
    };
    Cursor cur;
    cur.chunk = chunk0; cur.q = q0;
    if (cur.chunk >= nchunks || unit_of(cur.chunk, cur.q) >= nunits) return;
    cur.valid = true;
    cur.D = make_unit(unit_of(cur.chunk, cur.q));

    // ---- the prefetch side: one ring turn ahead of the compute side ------------------------------------------
    Cursor pre = cur;
    int pre_h = 0;            // next piece of pre.D to request
    uint32_t vb[4];           // per row group of 16: this lane's byte offset from the unit's base (row clamped to a valid one)
    UnitD PD = pre.D;         // the unit the requests go to (stays at the wave's last unit once `pre` has run off the end)
    auto set_vb = [&]() {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int row = RPI * g + rr;
            row = row < PD.rows ? row : PD.rows - 1;
            vb[g] = (uint32_t)row * pstride + (uint32_t)pp * 16u;
        }
    };
    set_vb();
    // a unit's request stream is padded to whole turns; after its last slot the stream moves to the wave's next unit
    auto pre_advance = [&]() {
        ++pre_h;
        const int padded = ((PD.nh + NSLOT - 1) / NSLOT) * NSLOT;
        if (pre_h >= padded && pre.valid) {
            next_unit(pre);
            if (pre.valid) { PD = pre.D; pre_h = 0; set_vb(); }
        }
    };
    uint4_t SR[IPS]; // STG 1: the slot in flight
    auto issue_regs = [&]() {
        int h = pre_h < PD.nh ? pre_h : PD.nh - 1;
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(h * SB);
        const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)PD.base_hi) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)PD.base_lo);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(base), 0, 0xffffffffu, 0x00020000);
        if (!(FL & 1) || __builtin_amdgcn_readfirstlane(pre_h) < __builtin_amdgcn_readfirstlane(PD.nh)) {
#pragma unroll
            for (int i = 0; i < IPS; ++i) {
                const int g = i / HPS, hh = i % HPS;
                SR[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vb[g], (int)(soff + (uint32_t)hh * 64u), 0);
            }
        }
        pre_advance();
    };
    auto land_regs = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        char *dst = ring + (STG ? 0 : s * SLOTB) + lane * 16;
#pragma unroll
        for (int i = 0; i < IPS; ++i) *reinterpret_cast<uint4_t *>(dst + i * 1024) = SR[i];
    };
    auto issue_slot = [&](auto sc) { // the next piece of the prefetch stream into ring slot s
        constexpr int s = decltype(sc)::value;
        int h = pre_h < PD.nh ? pre_h : PD.nh - 1; // past the row's end: re-read its last piece (an L2 hit), keeps the count of requests fixed
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(h * SB);
        const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)PD.base_hi) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)PD.base_lo);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(base), 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int i = 0; i < IPS; ++i) { // the halves of a row group's line in consecutive instructions
            const int g = i / HPS, hh = i % HPS;
            const uint32_t m0v = ring_lds + (uint32_t)(s * SLOTB + i * 1024);
            const uint32_t so = soff + (uint32_t)hh * 64u;
            unsigned keep; // M0 belongs to the compiler: saved and restored inside the statement
            if (NT)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(m0v), "v"(vb[g]), "s"(rsrc), "s"(so) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(m0v), "v"(vb[g]), "s"(rsrc), "s"(so) : "memory");
        }
        pre_advance();
    };

    // ---- per-lane state of the current locus ---------------------------------------------------------------
    double q[NJ], cs[NJ], dd[NJ], xy[NJ];
    double mincov = INFINITY;
    int n_missing = 0;
    unsigned long long isum = 0, iwsum = 0;
    uint2_t cy0 = {0u, 0u}, cy1 = {0u, 0u}; // the last 16 bytes of the previous slot

    auto pool = [&](const uint2_t &a, const uint2_t &b, const uint2_t &d, int pi) {
        if (WORK == 0) {
            const unsigned long long s6 = (unsigned long long)a.x + a.y + b.x + b.y + d.x + d.y;
            isum += s6;
            iwsum += (unsigned long long)(pi + 1) * ((unsigned long long)a.x + 2ull * a.y + 3ull * b.x + 4ull * b.y + 5ull * d.x + 6ull * d.y);
            return;
        }
        const uint32_t c[NJ] = {a.x, a.y, b.x, b.y, d.y}; // remove Ns: column 4 is not in play
        double cd[NJ], f[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cd[j] = (double)c[j];
        double rs = cd[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j) rs = rs + cd[j];
        const bool rowok = rs != 0.0;
        const double rsd = fmax(rs, 1.0);
        const double rinv = recip_for_div(rsd);
#pragma unroll
        for (int j = 0; j < NJ; ++j) f[j] = div_by(cd[j], rsd, rinv);
        mincov = fmin(mincov, rs);
        n_missing += rowok ? 0 : 1;
        const double2 t = *reinterpret_cast<const double2 *>(tab + 2 * pi); // broadcast read
        const double wi = t.x, yi = t.y;
#pragma unroll
        for (int j = 0; j < NJ; ++j) q[j] = q[j] + f[j] * wi;
#pragma unroll
        for (int j = 0; j < NJ; ++j) cs[j] = cs[j] + f[j];
#pragma unroll
        for (int j = 0; j < NJ; ++j) dd[j] = fma(f[j], f[j], dd[j]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) xy[j] = fma(f[j], yi, xy[j]);
    };

    // one unit, pool phase PH = delta % 24 (compile-time): pool slot j of turn T starts at byte PH + 24 j of the turn
    auto run_unit = [&](auto phc) {
        constexpr int PH = decltype(phc)::value;
        const UnitD D = cur.D;
        const int a = D.delta / 24;               // pool slots of the first turn that lie in front of the row
        const int nturn = (D.nh + NSLOT - 1) / NSLOT;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { q[j] = 0.0; cs[j] = 0.0; dd[j] = 0.0; xy[j] = 0.0; }
        mincov = INFINITY; n_missing = 0; isum = 0; iwsum = 0;
        for (int T = 0; T < nturn; ++T) {
            const int pbase = PPT * T - a;        // pool index of slot 0 of this turn
            static_for<0, NSLOT>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if (STG) {
                    __builtin_amdgcn_wave_barrier();
                    land_regs(sc);   // the compiler waits for the registers
                    __builtin_amdgcn_wave_barrier();
                    issue_regs();    // in flight while this slot is computed
                } else {
                    // the piece of this slot has landed once at most (NSLOT - 1) slots' requests are outstanding
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSLOT - 1) * IPS) : "memory");
                }
                const char *slot = cell + (STG ? 0 : s * SLOTB);
                // pools whose LAST byte lies in this slot: j = -1 (started in the previous turn's last slot) .. PPT - 1
                static_for<0, PPT + 1>([&](auto jc) {
                    constexpr int j = decltype(jc)::value - 1;
                    constexpr int u = PH + 24 * j;              // first byte, relative to the turn
                    constexpr int e = u + 23;                   // last byte
                    constexpr bool here = (e >= s * SB) && (e < (s + 1) * SB) && (j >= 0 || PH != 0);
                    // (j = -1 exists only when a pool straddles the turn boundary, i.e. PH != 0; its tail ends in slot 0)
                    if constexpr (here && (j >= 0 || s == 0)) {
                        // a pool of the previous turn that ends in slot 0 of this one carries index pbase - 1 ... in general pbase + j
                        const int pi = pbase + j;
                        if (pi >= 0 && pi < n) {
                            uint2_t wv[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const int uw = u + 8 * k; // (constexpr in effect: u and k are)
                                if (uw >= s * SB) wv[k] = *reinterpret_cast<const uint2_t *>(slot + ((uw - s * SB) / 64) * 1024 + (((uw - s * SB) / 16) & 3) * 256 + (uw & 8));
                                else wv[k] = (uw - (s * SB - 16)) == 0 ? cy0 : cy1; // one of the two carried words
                            }
                            pool(wv[0], wv[1], wv[2], pi);
                        }
                    }
                });
                // the last 16 bytes of this slot, for a pool that starts here and ends in the next slot
                {
                    const uint4_t tl = *reinterpret_cast<const uint4_t *>(slot + (HPS - 1) * 1024 + 3 * 256);
                    cy0 = uint2_t{tl.x, tl.y};
                    cy1 = uint2_t{tl.z, tl.w};
                }
                if (!STG) {
                    // every read of this slot has returned before its refill is requested
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue_slot(sc);
                }
            });
        }
        // "closing": one line of output per locus
        const int64_t l = D.r0 + (int64_t)lane * period;
        if ((FL & 2) && lane < D.rows && l < L) { // the operators' output arrays, as k_locus_close lays them out
            double s1 = mincov + (double)n_missing, s2 = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) { s1 += q[j] + cs[j]; s2 += dd[j] + xy[j]; }
            int32_t *o_n = reinterpret_cast<int32_t *>(out + 2 * L);
            int32_t *o_ids = o_n + L;
            double *o_mf = reinterpret_cast<double *>(o_ids + 5 * L + (L & 1));
            double *o_st = o_mf + 5 * L, *o_pv = o_st + 5 * L;
            o_n[l] = 1;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                o_ids[5 * l + r] = r == 0 ? 1 : -1;
                o_mf[5 * l + r] = r == 0 ? s1 : NAN;
                o_st[5 * l + r] = r == 0 ? s2 : NAN;
                o_pv[5 * l + r] = r == 0 ? s1 * s2 : NAN;
            }
        } else if (lane < D.rows && l < L) {
            if (WORK == 0) {
                out[2 * l] = (double)isum;
                out[2 * l + 1] = (double)iwsum;
            } else {
                double s1 = mincov + (double)n_missing, s2 = 0.0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { s1 += q[j] + cs[j]; s2 += dd[j] + xy[j]; }
                out[2 * l] = s1;
                out[2 * l + 1] = s2;
            }
        }
    };

    // prologue: the first turn of the first unit (STG 1: its first slot)
    if (STG) issue_regs();
    else static_for<0, NSLOT>([&](auto sc) { issue_slot(sc); });
    while (cur.valid) {
        const int ph = cur.D.delta % 24;
        if (ph == 0) run_unit(std::integral_constant<int, 0>{});
        else if (ph == 8) run_unit(std::integral_constant<int, 8>{});
        else run_unit(std::integral_constant<int, 16>{});
        next_unit(cur);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the dummy requests of the last turn
}

template <typename K, typename... A>
static float timeit(K kern, int grid, int threads, size_t shmem, A... args) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shmem, 0, args...);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

template <int SB, int NSLOT, int WORK, int WPB, int MINW, int NT = 0, int STG = 0, int FL = 0>
static int run(const char *tag, const uint32_t *counts, const double *wy, int64_t L, int n, int period, double *out, int cus, int blocks_per_cu,
               const std::vector<uint32_t> *host_counts) {
    auto kern = k_stream<SB, NSLOT, WORK, WPB, MINW, NT, STG, FL>;
    const size_t shmem = (size_t)WPB * (STG ? 1 : NSLOT) * 64 * SB + (size_t)n * 16;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int grid = cus * blocks_per_cu;
    CK(hipMemset(out, 0, sizeof(double) * 2 * L));
    const float ms = timeit(kern, grid, 64 * WPB, shmem, counts, wy, L, n, period, out);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    const double gb = 24.0 * n * (double)L;
    printf("%-34s SB=%d NSLOT=%d WORK=%d NT=%d STG=%d FL=%d waves/CU=%2d lds/blk=%6zu: %.3f ms  %.2f TB/s  frac %.3f", tag, SB, NSLOT, WORK, NT, STG, FL, WPB * blocks_per_cu, shmem,
           ms, gb / ms / 1e9, gb / ms / 1e9 / 8.0);
    if (WORK == 0 && host_counts) { // validate the plumbing: every row's sum and position-weighted sum
        std::vector<double> h((size_t)2 * L);
        CK(hipMemcpy(h.data(), out, sizeof(double) * 2 * L, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        const int64_t step = L > 4096 ? L / 2048 : 1;
        for (int64_t l = 0; l < L; l += (l < 256 || l > L - 256) ? 1 : step) {
            unsigned long long s = 0, ws = 0;
            for (int i = 0; i < n; ++i) {
                const uint32_t *c = host_counts->data() + ((size_t)l * n + i) * 6;
                s += (unsigned long long)c[0] + c[1] + c[2] + c[3] + c[4] + c[5];
                ws += (unsigned long long)(i + 1) * ((unsigned long long)c[0] + 2ull * c[1] + 3ull * c[2] + 4ull * c[3] + 5ull * c[4] + 6ull * c[5]);
            }
            if (h[2 * l] != (double)s || h[2 * l + 1] != (double)ws) { if (bad < 5) printf("\n  row %lld: got %.0f %.0f want %llu %llu", (long long)l, h[2 * l], h[2 * l + 1], s, ws); ++bad; }
        }
        printf("  check: %s", bad ? "FAILED" : "ok");
    }
    printf("\n");
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int n = argc > 1 ? atoi(argv[1]) : 100;
    const int64_t L = argc > 2 ? atoll(argv[2]) : 1000000;
    int period = 1;
    while ((((int64_t)n * 24 * period) & 127) != 0) period *= 2;
    printf("pools %d loci %lld period %d CUs %d\n", n, (long long)L, period, cus);
    std::vector<uint32_t> hc((size_t)L * n * 6);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < hc.size(); ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int col = (int)(i % 6);
        hc[i] = col < 2 ? (uint32_t)(x % 70) + 1 : (col == 5 ? (uint32_t)((x >> 20) % 50 == 0) : 0u); // A, T; a rare D
    }
    std::vector<double> hw((size_t)2 * n);
    for (int i = 0; i < n; ++i) { hw[2 * i] = 1.0 / n; hw[2 * i + 1] = std::sin(0.7 * i); }
    uint32_t *counts; double *wy, *out;
    CK(hipMalloc(&counts, hc.size() * 4 + 4096));
    CK(hipMalloc(&wy, hw.size() * 8));
    CK(hipMalloc(&out, sizeof(double) * 2 * L + (size_t)L * 160 + 64));
    CK(hipMemcpy(counts, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wy, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
    // plumbing first
    run<128, 3, 0, 4, 2, 0, 1>("check", counts, wy, L, n, period, out, cus, 2, &hc);
    run<128, 3, 0, 4, 2, 0, 1, 1>("check, no dummy requests", counts, wy, L, n, period, out, cus, 2, &hc);
    run<128, 3, 0, 4, 2, 0, 1, 5>("check, no dummies, wave groups", counts, wy, L, n, period, out, cus, 2, &hc);
    for (int rep = 0; rep < 2; ++rep) {
        run<128, 3, 0, 4, 2, 0, 1>("loads only", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 0, 4, 2, 0, 1, 1>("loads only", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 0, 4, 2, 0, 1, 5>("loads only", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 1, 4, 2, 0, 1>("filter+diag", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 1, 4, 2, 0, 1, 1>("filter+diag", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 1, 4, 2, 0, 1, 5>("filter+diag", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 1, 4, 2, 0, 1, 3>("filter+diag+outputs", counts, wy, L, n, period, out, cus, 2, nullptr);
        run<128, 3, 1, 4, 2, 0, 1, 7>("filter+diag+outputs", counts, wy, L, n, period, out, cus, 2, nullptr);
    }
    return 0;
}
