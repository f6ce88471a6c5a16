// rom_wide.hip -- the whole POD-PROM time loop of one sample on one compute unit for the thesis' LARGER bases, 40 < r <= 96
// (bg_rom_run_wide).  reference: FEMBurgers.pod_prom_burgers, FEM/fem_burgers.py:709-785; the bases are
// POD/modes/U_modes_tol_1e-04.npy (r = 96) and smaller, driven by POD/Results_thesis/prom_pod.py:35-58.
//
// bg_rom_run (rom_fused.hip) keeps the basis in registers: 2 N r bytes per lane-set, which ends at r = 40.  Here the basis
// streams through LDS, 64 mesh rows at a time (one pass over Phi per Picard iteration, L2-resident: 393 KB at r = 96), and
// what stays in registers are the ACCUMULATORS of the reduced system: its 4 x 4 block pairs are dealt round-robin to the
// four waves (LSPG 348 pairs: 87 per wave; Galerkin 600: 150 per wave -- one wave per SIMD, 512 registers), every wave
// sweeps ALL mesh rows for its own pairs, so there are no per-wave partial systems to add up.  The slabs are double
// buffered and arrive by LDS DMA (global_load_lds: no registers, the next slab lands while this one is worked on).  Per slab:
//   four lanes per row lift u = Phi q for rows i - 1, i, i + 1 (:773; iterations after the first) and assemble A(u), R(u)
//   of row i (:730-753, same arithmetic as every other kernel)  ->  each wave forms the rows of Y = A Phi it multiplies,
//   from the slab and the coefficients in LDS (lane (k, blk, t): mesh row 4 k + blk of the step, columns 24 t + c for
//   block c: twelve 16-byte reads per row and 24 blocks)  ->  v_mfma_f64_4x4x4_4b.  Two workgroup barriers per slab.
// Then the 96 x 96 solve by all four waves -- bg_rom_run's guarded pivot-free Gauss-Jordan on two row tiles (rows 0-63 and
// 64-95 of a column live in two registers of lane = row mod 64; wave w owns the 4-column blocks b = w mod 4), one panel
// of four columns at a time -- q = Phi^T u + dq, the stopping test, and after the last iteration one lift-only sweep for
// U[:, n+1] = Phi q (:779).  A sample whose elimination meets a multiplier above 1 (np.linalg.solve would have exchanged rows)
// is marked BG_INFO_NEEDS_PIVOTING and redone by the caller through the library path (burgers_hip/rom.py does).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "rom_device.hpp"

namespace {

using namespace bg;

constexpr int WR = 96;                 // padded reduced dimension: column 24 t + c  <->  (lane index t, block c)
constexpr int WNB = 24;                // 4-column blocks
constexpr int WRS = 64;                // mesh rows per slab
constexpr int WPS = 98;                // doubles per row of the LDS slabs and of the parked system (16-byte aligned rows)
constexpr int WSLAB = (WRS + 2) * WPS;                       // doubles of one slab buffer: mesh rows [r0 - 1, r0 + 64]
constexpr int WCHUNKS = (WSLAB * 8 + 1023) / 1024;           // 1-KB LDS DMA pieces per slab (51)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

struct WideRunArgs {
    const double* x;        // [N]
    const double* PhiP;     // [NPAD + 2][96]: Phi row i at index i + 1, zero rows around and beyond N, zero columns beyond r
    const double* u0;       // [B][N]
    const double* mu1;      // [B]
    const double* mu2;      // [B]
    double* hist;           // [B][nsteps+1][N]
    int32_t* iters;         // [B][nsteps]
    int32_t* flags;         // [B]
    int32_t* info;          // [B]
    const int32_t* order;   // [B] or null: slot i of the persistent loop works on sample order[i]
    double dt, E, tol;
    int N, NPAD, B, r, nsteps, max_it, supg, nonuniform, force_handback;
};

template <bool GAL>
struct WideItems {
    // LSPG: pairs (ca <= cb) of Y, then (Y[ca], X) for br, then (Phi[ca], X) for Phi^T u; Galerkin: (Phi[ca], Y[cb]), then (Phi[ca], X)
    static constexpr int pairs = GAL ? WNB * WNB : WNB * (WNB + 1) / 2;
    static constexpr int total = pairs + (GAL ? WNB : 2 * WNB);
    static constexpr int per_wave = (total + 3) / 4;
};

// The matrix instructions of one 16-row step for wave W: item i of the fixed enumeration belongs to wave i % 4, accumulator i / 4.
template <bool GAL, int W>
__device__ __forceinline__ void wide_step_mfma(const double (&Y)[WNB], const double (&P)[WNB], double X, double (&acc)[WideItems<GAL>::per_wave])
{
    int i = 0;
    if constexpr (GAL) {
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca) {
#pragma unroll
            for (int cb = 0; cb < WNB; ++cb, ++i)
                if (i % 4 == W) acc[i / 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(P[ca], Y[cb], acc[i / 4], 0, 0, 0);
        }
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca, ++i)
            if (i % 4 == W) acc[i / 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(P[ca], X, acc[i / 4], 0, 0, 0);
    } else {
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca) {
#pragma unroll
            for (int cb = ca; cb < WNB; ++cb, ++i)
                if (i % 4 == W) acc[i / 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(Y[ca], Y[cb], acc[i / 4], 0, 0, 0);
        }
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca, ++i)
            if (i % 4 == W) acc[i / 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(Y[ca], X, acc[i / 4], 0, 0, 0);
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca, ++i)
            if (i % 4 == W) acc[i / 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(P[ca], X, acc[i / 4], 0, 0, 0);
    }
}

// Sum the four block partials of wave W's accumulators and park them: S[24 i + ca][24 j + cb] = Ar, column 96 = br, 97 = Phi^T u.
template <bool GAL, int W>
__device__ __forceinline__ void wide_park(const double (&acc)[WideItems<GAL>::per_wave], double* __restrict__ S, int lane)
{
    const int oi = lane >> 4, oj = lane & 3;
    const bool writer = ((lane >> 2) & 3) == 3;
    auto put = [&](int i, int row_c, int col_c, int kind) {      // kind 0: block pair, 1: br (column j = 0), 2: Phi^T u (column j = 1)
        if (i % 4 != W) return;
        double v = acc[i / 4];
        v += dpp_mov<0x114>(v);              // row_shr:4
        v += dpp_mov<0x118>(v);              // row_shr:8 -> lanes with blk == 3 hold the sum
        if (kind == 0) { if (writer) S[(24 * oi + row_c) * WPS + 24 * oj + col_c] = v; }
        else if (kind == 1) { if (writer && oj == 0) S[(24 * oi + row_c) * WPS + WR] = v; }
        else { if (writer && oj == 1) S[(24 * oi + row_c) * WPS + WR + 1] = v; }
    };
    int i = 0;
    if constexpr (GAL) {
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca)
#pragma unroll
            for (int cb = 0; cb < WNB; ++cb, ++i) put(i, ca, cb, 0);
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca, ++i) { put(i, ca, 0, 1); put(i, ca, 0, 2); }
    } else {
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca)
#pragma unroll
            for (int cb = ca; cb < WNB; ++cb, ++i) put(i, ca, cb, 0);
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca, ++i) put(i, ca, 0, 1);
#pragma unroll
        for (int ca = 0; ca < WNB; ++ca, ++i) put(i, ca, 0, 2);
    }
}

typedef __attribute__((address_space(3))) double lds_double_t;
typedef __attribute__((address_space(3))) int lds_int_t;

// The 96 x 96 solve of one iteration by all four waves: ONE out-of-line copy for every wave and both halves of the panel
// range (the wave number and the panel are run-time values here).  Inlined into the four per-wave bodies and unrolled over
// its 24 panels it was 160 KB of straight-line code per pass and CU -- four waves streaming four different copies through
// the instruction cache -- and took longer than the projection (111 k of 218 k clocks per pass).
// In: the parked system S (Ar | br, LSPG: upper blocks).  Out: s_diag, s_y (x_k = y_k / d_k), s_bad[w] = guard of this wave.
// wave w owns the column blocks b = w, w + 4, ... (six of 24); the right-hand side rides with wave 3.
// col[s][tile][tt] = entry (row 64 tile + lane, column 4 (w + 4 s) + tt).  Contains workgroup barriers: all waves call it.
template <bool GAL>
__device__ __attribute__((noinline)) void wide_solve(const lds_double_t* S, lds_double_t* s_m, lds_double_t* s_diag, lds_double_t* s_y,
                                                     lds_int_t* s_bad, int w, int lane, int r)
{
    double col[6][2][4], rhs[2];
    auto entry = [&](int i, int j) -> double {   // (Ar | br)[i][j]; LSPG: the lower blocks by symmetry
        int rr = i, cc = j;
        if (!GAL && j < WR && (i % 24) > (j % 24)) { rr = j; cc = i; }
        return S[rr * WPS + cc];
    };
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        const int row = 64 * tile + lane;
        const bool rin = row < WR;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int j = 4 * (w + 4 * s) + tt;
                double v = 0.0;
                if (rin) v = (row >= r || j >= r) ? ((row == j) ? 1.0 : 0.0) : entry(row, j);
                col[s][tile][tt] = v;
            }
        }
        rhs[tile] = (w == 3 && rin && row < r) ? -entry(row, WR) : 0.0;
    }
    double gmax = 0.0;
    bool zero_piv = false;
    // factor the panel p held in slot OS (pivot rows in row tile TK); multipliers of all 96 rows -> s_m[p & 1][kk][row]
    auto factor = [&](int p, auto os_c, auto tk_c) {
        constexpr int OS = decltype(os_c)::value, TK = decltype(tk_c)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kr = 4 * p + kk, lk = kr & 63;
            const double piv = readlane_f64(col[OS][TK][kk], lk);
            const double rp = rcp(piv);
            zero_piv = zero_piv | (piv == 0.0);
            double pvj[4];
#pragma unroll
            for (int jj = kk + 1; jj < 4; ++jj) pvj[jj] = readlane_f64(col[OS][TK][jj], lk);
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
                const int row = 64 * tile + lane;
                const double m = (row != kr && row < WR) ? col[OS][tile][kk] * rp : 0.0;    // rows above the pivot too
                gmax = fmax(gmax, (row > kr) ? fabs(m) : 0.0);
#pragma unroll
                for (int jj = kk + 1; jj < 4; ++jj) col[OS][tile][jj] = __builtin_fma(-m, pvj[jj], col[OS][tile][jj]);
                if (row < WR) s_m[((p & 1) * 4 + kk) * WR + row] = m;
            }
        }
    };
    // apply panel p (multipliers m, pivot rows in row tile TK) to the block in slot SL
    auto apply = [&](int p, auto sl_c, auto tk_c, const double (&m)[4][2]) {
        constexpr int SL = decltype(sl_c)::value, TK = decltype(tk_c)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int lk = (4 * p + kk) & 63;
            double pv[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) pv[tt] = readlane_f64(col[SL][TK][tt], lk);
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) col[SL][tile][tt] = __builtin_fma(-m[kk][tile], pv[tt], col[SL][tile][tt]);
        }
    };
    // `slot` is a run-time value: pick the register block with a (wave-uniform) switch
    auto with_slot = [&](int slot, auto&& f) {
        switch (slot) {
            case 0: f(std::integral_constant<int, 0>{}); break;
            case 1: f(std::integral_constant<int, 1>{}); break;
            case 2: f(std::integral_constant<int, 2>{}); break;
            case 3: f(std::integral_constant<int, 3>{}); break;
            case 4: f(std::integral_constant<int, 4>{}); break;
            default: f(std::integral_constant<int, 5>{}); break;
        }
    };
    __syncthreads();                               // every wave has its columns: the parked system is dead
    if (w == 0) factor(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    __syncthreads();
    // One barrier per panel: while the other waves apply panel p to their later blocks, the owner of panel p + 1 brings that
    // block up to date first, factors it and publishes its multipliers (look-ahead), then does the rest.
    double mprev[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    auto panels = [&](int p0, int p1, auto tk_c, auto tkprev_c) {   // pivots of panels [p0, p1) in row tile TK, of panel p0 - 1 in TKPREV
#pragma unroll 1
        for (int p = p0; p < p1; ++p) {
            double m[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                m[kk][0] = s_m[((p & 1) * 4 + kk) * WR + lane];
                m[kk][1] = (lane < WR - 64) ? s_m[((p & 1) * 4 + kk) * WR + 64 + lane] : 0.0;
            }
            const int nxt = p + 1;
            const bool owner_next = nxt < WNB && (nxt & 3) == w;
            const bool owner_this = p > 0 && (p & 3) == w;       // owned panel p: its other blocks still lack panel p - 1
            if (owner_next) {
                with_slot(nxt >> 2, [&](auto sl) {
                    apply(p, sl, tk_c, m);
                    if (nxt < 16) factor(nxt, sl, std::integral_constant<int, 0>{});
                    else factor(nxt, sl, std::integral_constant<int, 1>{});
                });
            } else {
                // the wave that factors the next panel leaves its other blocks for the next round (it is on the critical
                // path: one block update + one factorisation against six block updates of the others)
                auto own = [&](auto sl) {                  // this wave's block in slot SL
                    const int b = w + 4 * decltype(sl)::value;
                    if (b > p) {
                        if (owner_this) apply(p - 1, sl, tkprev_c, mprev);
                        apply(p, sl, tk_c, m);
                    }
                };
                own(std::integral_constant<int, 0>{}); own(std::integral_constant<int, 1>{}); own(std::integral_constant<int, 2>{});
                own(std::integral_constant<int, 3>{}); own(std::integral_constant<int, 4>{}); own(std::integral_constant<int, 5>{});
            }
            if (w == 3) {
                constexpr int TK = decltype(tk_c)::value;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int lk = (4 * p + kk) & 63;
                    const double pv = readlane_f64(rhs[TK], lk);
                    rhs[0] = __builtin_fma(-m[kk][0], pv, rhs[0]);
                    rhs[1] = __builtin_fma(-m[kk][1], pv, rhs[1]);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { mprev[kk][0] = m[kk][0]; mprev[kk][1] = m[kk][1]; }
            if (p + 1 < WNB) __syncthreads();
        }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    panels(0, 16, T0{}, T0{});
    panels(16, 17, T1{}, T0{});                    // panel 16's deferred predecessor (15) has its pivots in the first row tile
    panels(17, WNB, T1{}, T1{});
    // what is left is diagonal: x_k = y_k / d_k.  Publish d (the owner of each column) and y (wave 3).
#pragma unroll
    for (int s = 0; s < 6; ++s) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int j = 4 * (w + 4 * s) + tt;   // this column's diagonal entry sits in row j
            if (lane == (j & 63)) s_diag[j] = (j >> 6) ? col[s][1][tt] : col[s][0][tt];
        }
    }
    if (w == 3) {
        s_y[lane] = rhs[0];
        if (lane < WR - 64) s_y[64 + lane] = rhs[1];
    }
    {
        const unsigned long long anybad = __ballot(zero_piv | !(gmax <= 1.0));
        if (lane == 0) s_bad[w] = anybad != 0ull;
    }
    __syncthreads();
}

#ifdef BG_WIDE_TIMING                   // diagnostic builds (tools/time_wide_rom.py --phases): kilo-clocks per phase in place of the counts
constexpr bool kWT = true;
#else
constexpr bool kWT = false;
#endif

struct WideLdsPtrs {
    double* slab;           // two slab buffers; later the system
    double* u;              // [NPADM + 4]
    double* g; double* h; double* fdt;      // [NPADM]
    double (*cf)[4];        // [NPADM][4]
    double* q;              // [WR]
    double (*m)[WR];        // [2][4][WR]: multipliers of the current and the next panel
    double* diag; double* y;
    int* bad;
};

// The body of the kernel for wave W of the workgroup.  The wave number is a template parameter of the WHOLE body (the kernel
// branches once, at its top): every wave runs its own quarter of the block pairs with accumulators that never change
// registers.  A `switch (w)` around the matrix instructions of each row step instead cost 340 accumulator moves per
// 87 instructions (first version: 6.5e5 sample-steps/s).  All four copies execute the same sequence of barriers.
template <bool GAL, int W>
__device__ __forceinline__ void rom_wide_body(const WideRunArgs& a, const WideLdsPtrs& L)
{
    constexpr int NPADM = 512;
    constexpr int NACC = WideItems<GAL>::per_wave;
    constexpr int w = W;
    double* const s_slab = L.slab;
    double* const s_u = L.u;
    double* const s_g = L.g;
    double* const s_h = L.h;
    double* const s_fdt = L.fdt;
    double (*const s_cf)[4] = L.cf;
    double* const s_q = L.q;
    double (*const s_m)[WR] = L.m;
    double* const s_diag = L.diag;
    double* const s_y = L.y;
    int* const s_bad = L.bad;
    double* const S = s_slab;                                    // [WR][WPS]: Ar | br | Phi^T u (over the dead slabs)

    const int tid = threadIdx.x;
    const int N = a.N, r = a.r;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const int nslab = (N + WRS - 1) / WRS;
    if (tid < 4) s_u[tid < 2 ? tid : NPADM + tid] = 0.0;

    for (int slot = blockIdx.x; slot < a.B; slot += gridDim.x) {
        const int smp = a.order ? a.order[slot] : slot;
        const double mu1 = a.mu1[smp], mu2 = a.mu2[smp];
        double* hist = a.hist + (size_t)smp * (size_t)(a.nsteps + 1) * (size_t)N;
        __syncthreads();
        // ---- per-sample constants (compute_forcing_vector :427-461, f_gp of :556-558) and the initial state ------------
        for (int i = tid; i < NPADM; i += 256) {
            double frPrev = 0.0, fl = 0.0, hf = 0.0, u = 0.0;
            if (i < N) {
                if (i > 0) {
                    const double xl = a.x[i - 1], xr = a.x[i];
                    const double he = a.nonuniform ? xr - xl : h;
                    const double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
                    const double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
                    frPrev = (f1 * GP_B + f2 * GP_A) * (0.5 * he);
                }
                if (i < N - 1) {
                    const double xl = a.x[i], xr = a.x[i + 1];
                    const double he = a.nonuniform ? xr - xl : h;
                    const double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
                    const double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
                    fl = (f1 * GP_A + f2 * GP_B) * (0.5 * he);
                    hf = he * (f1 + f2);
                }
                u = a.u0[(size_t)smp * N + i];
                hist[i] = u;
            }
            s_fdt[i] = a.dt * (frPrev + fl);
            s_h[i] = hf;
            s_u[i + 2] = u;
        }
        if (tid < WR) s_q[tid] = 0.0;
        __syncthreads();

        int flags = 0, info_out = 0;
        bool aborted = false;
        long long cyc[6] = {0, 0, 0, 0, 0, 0};
        long long tick = kWT ? (long long)__builtin_amdgcn_s_memtime() : 0;
        int npass = 0;
        auto lap = [&](int i) {
            if constexpr (kWT) {
                const long long now = (long long)__builtin_amdgcn_s_memtime();
                cyc[i] += now - tick;
                tick = now;
            }
        };
        // LDS DMA of slab `slab` (mesh rows [r0 - 1, r0 + 64] = rows r0 .. r0 + 65 of PhiP) into buffer `buf`: wave w moves the
        // 1-KB pieces w, w + 4, ...; a lane's 16 bytes land at piece base + 16 lane, i.e. LDS row o / 784, byte o % 784 of it
        // (the 16 bytes of row padding are filled from a valid dummy address)
        auto slab_dma = [&](int slab, int buf) {
            const char* src = reinterpret_cast<const char*>(a.PhiP + (size_t)slab * WRS * WR);
            const int ln = tid & 63;
            for (int j = w; j < WCHUNKS; j += 4) {
                const int o = 1024 * j + 16 * ln;
                const int row = o / (WPS * 8), within = o - row * (WPS * 8);
                const char* g = src + (within < WR * 8 ? row * (WR * 8) + within : 0);
                if (row < WRS + 2)                           // lanes beyond the slab's last row write nothing (the next buffer starts there)
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(reinterpret_cast<char*>(s_slab + buf * WSLAB) + 1024 * j), 16, 0, 0);
            }
        };

        for (int step = 0; step < a.nsteps && info_out == 0 && !aborted; ++step) {
            // ---- g = M u^n + dt F (`M @ U[:, n] + At*F`, :746) -----------------------------------------------------------
            for (int i = tid; i < NPADM; i += 256) {
                double g = 0.0;
                if (i < N) {
                    const double um = s_u[i + 1], u0 = s_u[i + 2], ur = s_u[i + 3];
                    if (a.nonuniform) {
                        double v = 0.0;
                        if (i > 0) v = (a.x[i] - a.x[i - 1]) / 6.0 * __builtin_fma(2.0, u0, um);
                        if (i < N - 1) v = __builtin_fma((a.x[i + 1] - a.x[i]) / 6.0, __builtin_fma(2.0, u0, ur), v);
                        g = v + s_fdt[i];
                    } else {
                        double acc;
                        if (i == 0) acc = __builtin_fma(2.0, u0, ur);
                        else if (i == N - 1) acc = __builtin_fma(2.0, u0, um);
                        else acc = __builtin_fma(4.0, u0, um) + ur;
                        g = __builtin_fma(h / 6.0, acc, s_fdt[i]);
                    }
                }
                s_g[i] = g;
            }
            __syncthreads();
            int k = 0;
            bool proj = true;
            while (true) {
                // per-lane indices from an opaque copy of the thread index: their address arithmetic is recomputed per pass
                // instead of being hoisted out of the time loop and spilled (see rom_fused.hip)
                int tid_i = tid;
                asm volatile("" : "+v"(tid_i));
                const int lane = tid_i & 63, pk = lane >> 4, pblk = (lane >> 2) & 3, pt = lane & 3;
                const bool lift = k > 0;                 // iteration 0 of a step assembles at u^n, which s_u holds (:725)
                double acc[NACC];
#pragma unroll
                for (int p = 0; p < NACC; ++p) acc[p] = 0.0;
                ++npass;
                lap(5);
                slab_dma(0, 0);                                // (not across the pass boundary: the parked system lies over both buffers)
                for (int slab = 0; slab < nslab; ++slab) {
                    const int r0 = slab * WRS, cur = slab & 1;
                    const double* s_P = s_slab + cur * WSLAB;                 // local row l = mesh row r0 - 1 + l
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces of the slab have landed
                    __syncthreads();                                          // ... and everybody's; the other buffer is no longer read
                    lap(0);
                    if (slab + 1 < nslab) slab_dma(slab + 1, cur ^ 1);        // the next slab lands while this one is worked on
                    // ---- four lanes per row i = r0 + q4: u_{i-1}, u_i, u_{i+1} = Phi q (:773), then A(u), R(u) of row i --------------
                    {
                        const int q4 = tid_i >> 2, i = r0 + q4;
                        double um, u0, ur;
                        if (lift) {
                            const double* prow = s_P + q4 * WPS + 24 * pt;
                            double sm = 0.0, s0 = 0.0, sr = 0.0;
#pragma unroll
                            for (int c2 = 0; c2 < 12; ++c2) {
                                const double2 qv = *reinterpret_cast<const double2*>(&s_q[24 * pt + 2 * c2]);
                                const double2 pm = *reinterpret_cast<const double2*>(prow + 2 * c2);
                                const double2 p0 = *reinterpret_cast<const double2*>(prow + WPS + 2 * c2);
                                const double2 pr = *reinterpret_cast<const double2*>(prow + 2 * WPS + 2 * c2);
                                sm = __builtin_fma(pm.x, qv.x, sm); sm = __builtin_fma(pm.y, qv.y, sm);
                                s0 = __builtin_fma(p0.x, qv.x, s0); s0 = __builtin_fma(p0.y, qv.y, s0);
                                sr = __builtin_fma(pr.x, qv.x, sr); sr = __builtin_fma(pr.y, qv.y, sr);
                            }
                            sm += dpp_mov<0xB1>(sm); sm += dpp_mov<0x4E>(sm);          // quad sums: every lane of the quad holds the three values
                            s0 += dpp_mov<0xB1>(s0); s0 += dpp_mov<0x4E>(s0);
                            sr += dpp_mov<0xB1>(sr); sr += dpp_mov<0x4E>(sr);
                            um = sm; u0 = s0; ur = sr;                                 // rows outside the mesh are zero rows of PhiP
                            if (pt == 0) s_u[i + 2] = u0;
                        } else {
                            um = s_u[i + 1]; u0 = s_u[i + 2]; ur = s_u[i + 3];
                        }
                        if (proj && pt == 0) {
                            const bool in = i < N;
                            const MeshConst mc = make_mesh_const(h, a.dt, a.E, a.supg);
                            double lo, di, up, R;
                            rom_assemble_row(i, N, um, u0, (i + 1 < N) ? ur : 0.0, in ? s_g[i] : 0.0,
                                             (in && i > 0) ? s_h[i - 1] : 0.0, (in && i < N - 1) ? s_h[i] : 0.0, mu1, mc,
                                             a.nonuniform, a.x, a.dt, a.E, lo, di, up, R);
                            *reinterpret_cast<double2*>(&s_cf[i][0]) = make_double2(lo, di);
                            *reinterpret_cast<double2*>(&s_cf[i][2]) = make_double2(up, R);
                        }
                    }
                    lap(1);
                    if (proj) {
                        __syncthreads();                                      // the slab's coefficients (and u) are in LDS
                        // ---- projection: four steps of 16 rows; lane (k, blk, t): row 16 st + 4 k + blk, columns 24 t + c -------------
#pragma unroll 1
                        for (int st = 0; st < WRS / 16; ++st) {
                            const int rl = 16 * st + 4 * pk + pblk;
                            const double2 c01 = *reinterpret_cast<const double2*>(&s_cf[r0 + rl][0]);
                            const double2 c23 = *reinterpret_cast<const double2*>(&s_cf[r0 + rl][2]);
                            const double* pb = s_P + rl * WPS + 24 * pt;          // the row below (local row rl = mesh row r0 - 1 + rl)
                            double Y[WNB], P[WNB];
#pragma unroll
                            for (int c2 = 0; c2 < 12; ++c2) {
                                const double2 tb = *reinterpret_cast<const double2*>(pb + 2 * c2);
                                const double2 tm = *reinterpret_cast<const double2*>(pb + WPS + 2 * c2);
                                const double2 ta = *reinterpret_cast<const double2*>(pb + 2 * WPS + 2 * c2);
                                P[2 * c2] = tm.x; P[2 * c2 + 1] = tm.y;
                                Y[2 * c2] = __builtin_fma(c23.x, ta.x, __builtin_fma(c01.y, tm.x, c01.x * tb.x));
                                Y[2 * c2 + 1] = __builtin_fma(c23.x, ta.y, __builtin_fma(c01.y, tm.y, c01.x * tb.y));
                            }
                            const double ui = s_u[r0 + rl + 2];
                            const double X = (pt == 0) ? c23.y : ((pt == 1) ? ui : 0.0);      // extra B block [R, u, 0, 0]
                            wide_step_mfma<GAL, W>(Y, P, X, acc);
                        }
                    }
                    lap(2);
                }
                __syncthreads();                               // the last slab's rows are no longer read (the system is parked over them)
                if (!proj) break;                              // that was the lift for U[:, n+1] = Phi q (:779)
                // ---- park the reduced system (over the dead slabs) ---------------------------------------------------------------
                wide_park<GAL, W>(acc, S, lane);
                __syncthreads();
                lap(3);
                // ---- solve(Ar, -br) (:767): guarded pivot-free Gauss-Jordan, two row tiles, panels of four columns ---------------
                const double wtu0 = (lane < r) ? S[lane * WPS + WR + 1] : 0.0;                     // Phi^T u, rows 0 .. 63
                const double wtu1 = (64 + lane < r) ? S[(64 + lane) * WPS + WR + 1] : 0.0;         // rows 64 .. 95
                wide_solve<GAL>((lds_double_t*)S, (lds_double_t*)&s_m[0][0], (lds_double_t*)s_diag, (lds_double_t*)s_y,
                                (lds_int_t*)s_bad, W, lane, r);
                lap(4);
                const bool tripped = ((s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) != 0) || a.force_handback;   // workgroup-uniform
                if (tripped) aborted = true;
                // ---- q = Phi^T u_k + dq, err = |dq| / |q|  (:770-776) -----------------------------------------------------------
                const double dq0 = (lane < r) ? s_y[lane] * rcp(s_diag[lane]) : 0.0;
                const double dq1 = (64 + lane < r) ? s_y[64 + lane] * rcp(s_diag[64 + lane]) : 0.0;
                const double q0 = wtu0 + dq0, q1 = wtu1 + dq1;
                double nd, nq;
                wave_sum2(dq0 * dq0 + dq1 * dq1, q0 * q0 + q1 * q1, nd, nq);
                nd = sqrt(nd); nq = sqrt(nq);
                const double err = nd / nq;
                ++k;
                const bool more = (err > a.tol) && (k < a.max_it) && !aborted;
                if (!(err - err == 0.0)) flags |= BG_FLAG_NONFINITE;
                if (k >= a.max_it) flags |= BG_FLAG_HIT_CAP;
                if (w == 0) {
                    s_q[lane] = q0;
                    if (lane < WR - 64) s_q[64 + lane] = q1;
                }
                __syncthreads();
                if (aborted) break;
                proj = more;                                   // after the last iteration: one lift-only sweep
            }
            // ---- U[:, n+1] = U1 (:779): one coalesced row ---------------------------------------------------------------------
            double* hrow = hist + (size_t)(step + 1) * N;
            for (int i = tid; i < N; i += 256) hrow[i] = s_u[i + 2];
            if (tid == 0) a.iters[(size_t)smp * a.nsteps + step] = k;
        }
        if (tid == 0) {
            a.flags[smp] = flags;
            a.info[smp] = aborted ? BG_INFO_NEEDS_PIVOTING : info_out;
        }
        if (kWT && W == 0 && tid == 0 && a.nsteps >= 8) {
            for (int i = 0; i < 6; ++i) a.iters[(size_t)smp * a.nsteps + i] = (int)(cyc[i] >> 10);
            a.iters[(size_t)smp * a.nsteps + 6] = npass;
        }
    }
}

template <bool GAL>
__global__ __launch_bounds__(256, 1) void rom_wide_kernel(WideRunArgs a)
{
    constexpr int NPADM = 512;
    __shared__ __attribute__((aligned(16))) double s_slab[2 * WSLAB + 128];      // two slab buffers (+ slack of the last DMA piece); later the system
    __shared__ __attribute__((aligned(16))) double s_u[NPADM + 4];                // u at offset 2, zero halo on each side
    __shared__ double s_g[NPADM], s_h[NPADM], s_fdt[NPADM];
    __shared__ __attribute__((aligned(16))) double s_cf[NPADM][4];                // lo, di, up, R per mesh row
    __shared__ __attribute__((aligned(16))) double s_q[WR];
    __shared__ double s_m[8][WR];                                                 // multipliers of the current and the next panel
    __shared__ double s_diag[WR], s_y[WR];
    __shared__ int s_bad[4];
    const WideLdsPtrs L{s_slab, s_u, s_g, s_h, s_fdt, s_cf, s_q, s_m, s_diag, s_y, s_bad};
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {       // wave-uniform by construction
        case 0: rom_wide_body<GAL, 0>(a, L); break;
        case 1: rom_wide_body<GAL, 1>(a, L); break;
        case 2: rom_wide_body<GAL, 2>(a, L); break;
        default: rom_wide_body<GAL, 3>(a, L); break;
    }
}

}  // namespace

extern "C" {

int bg_rom_run_wide_max_r(void) { return WR; }

// doubles of the padded basis copy bg_rom_run_wide reads: (NPAD + 2) rows of 96, NPAD = N rounded up to 64
long long bg_rom_run_wide_phi_elems(int N) { return N < 2 ? 0 : (long long)(((N + 63) / 64) * 64 + 2) * WR; }

int bg_rom_run_wide(int N, int B, int r, int nsteps, int projection, const double* x, const double* PhiP, const double* u0,
                    const double* mu1, const double* mu2, double dt, double E, double tol, int max_it, int options,
                    double* hist, int32_t* iters, int32_t* flags, int32_t* info, const int32_t* order, void* stream)
{
    if (N < 2 || B < 0 || r < 1 || nsteps < 0 || max_it < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (projection != BG_PROJ_GALERKIN && projection != BG_PROJ_LSPG) return BG_ERR_PROJECTION;
    if (N > 512) return BG_ERR_UNSUPPORTED_N;
    if (r > WR) return BG_ERR_UNSUPPORTED_R;
    if (B == 0) return BG_OK;
    if (!x || !PhiP || !u0 || !mu1 || !mu2 || !hist || !flags || !info || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    if ((uintptr_t)PhiP & 15) return BG_ERR_BAD_ARG;
    WideRunArgs a;
    a.x = x; a.PhiP = PhiP; a.u0 = u0; a.mu1 = mu1; a.mu2 = mu2; a.hist = hist; a.iters = iters; a.flags = flags; a.info = info; a.order = order;
    a.dt = dt; a.E = E; a.tol = tol; a.N = N; a.NPAD = ((N + 63) / 64) * 64; a.B = B; a.r = r; a.nsteps = nsteps; a.max_it = max_it;
    a.supg = options & BG_OPT_SUPG; a.nonuniform = (options & BG_OPT_NONUNIFORM) ? 1 : 0;
    a.force_handback = (options & BG_OPT_FORCE_PIVOTED) ? 1 : 0;      // tests: every sample is handed back to the caller
    const int cus = device_cu_count();
    const int grid = B < cus ? B : cus;
    hipStream_t st = (hipStream_t)stream;
    if (projection == BG_PROJ_GALERKIN)
        hipLaunchKernelGGL((rom_wide_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((rom_wide_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    return check_launch();
}

}  // extern "C"
