// rom_fused.hip -- the whole POD-PROM time loop of one sample on one compute unit, and its C-ABI entry point.
//
// Replaces FEMBurgers.pod_prom_burgers (reference FEM/fem_burgers.py:709-785) for a batch of samples: one 256-thread
// workgroup owns a sample for ALL time steps and ALL Picard iterations.  The basis fragments stay in registers (their halo
// rows in LDS), u, g, the per-sample load constants, the reduced system and q stay in LDS; HBM sees the initial state
// once and one N-row history write per time step.  Per iteration, with no kernel boundary and no host in between:
//     assembly (2 rows per thread)  ->  fp64 MFMA projection (v_mfma_f64_4x4x4_4b, as rom_reduce4_kernel)
//     ->  r x r solve by all four waves (below)  ->  q = Phi^T u + dq, error, stopping test  ->  lift u = Phi q.
//
// The reduced solve.  np.linalg.solve (:767) is LU with partial pivoting.  On these systems (cond(Ar) < 10 on the
// reference's bases) LAPACK never leaves the diagonal, so the kernel eliminates WITHOUT a pivot search and watches
// the multipliers: as long as every |l_ik| <= 1 the diagonal was the column maximum at every step, i.e. the
// operations are the ones partial pivoting performs.  If one multiplier exceeds 1 (or a pivot is 0) the sample is
// marked and redone from u0 by the repair kernel (PIV = true) launched behind the fast one: same code with the
// partial-pivoting single-wave routine of bg_lu_solve (lu_pivoted_wave) as its solve.
// The unpivoted elimination is spread over the four waves: lane = row, wave w owns the 4-column blocks
// b = w, w+4, w+8 (the right-hand side rides with wave 3); the owner of panel p factors it in its own registers
// (no communication inside a panel), publishes the four multiplier vectors through LDS, and every wave applies them
// to its own columns with v_readlane broadcasts of the pivot rows.  The owner of panel p+1 updates that block first
// and factors it while the others are still applying panel p (look-ahead): one barrier per panel.  Every update is an
// FMA over all 64 lanes, so the multipliers are formed for the rows ABOVE the pivot too (Gauss-Jordan) at no extra
// instruction: the elimination ends with a diagonal system and the 40 dependent steps of a back substitution vanish.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "rom_fused_device.hpp"

namespace {

using namespace bg;
using namespace bg::fused;

struct RomRunArgs {
    const double* x;        // [N]
    const double* Phi;      // [N][r]
    const double* u0;       // [B][N]
    const double* mu1;      // [B]
    const double* mu2;      // [B]
    double* hist;           // [B][nsteps+1][N]
    int32_t* iters;         // [B][nsteps]
    int32_t* flags;         // [B]
    int32_t* info;          // [B]
    const int32_t* order;   // [B] or null: slot i of the persistent loop works on sample order[i]
    double dt, E, tol;
    int N, B, r, nsteps, max_it, supg, nonuniform, force_pivoted;
};

// Two workgroups per CU (round 3): the fp64 MFMAs of a wave do not overlap with its own vector-ALU work
// (tools/mfma_valu_bench.hip), but the matrix pipe and the VALU of a SIMD serve different waves at the same time, so a
// second resident sample fills the phases in which the first one solves, lifts or waits at a barrier.  What that takes
// is LDS <= 80 KB per workgroup (was 135 KB): the per-thread halo table (40 KB) became register traffic between
// neighbouring lanes (HaloLanes), and the four per-wave partial systems (56 KB) two that wave pairs share (NRED = 2).
// The repair kernel (PIV) keeps one workgroup per CU: its one-wave pivoted solve holds a 41-double row per lane.
template <int S, int NB, int PROJ, bool PIV>
__global__ __launch_bounds__(256, PIV ? 1 : 2) void rom_fused_kernel(RomRunArgs a)
{
    constexpr int NPAD = 64 * S;
    constexpr int RW = 4 * NB;                   // padded reduced dimension
    constexpr bool GAL = PROJ == BG_PROJ_GALERKIN;
    __shared__ double s_u[NPAD + 4];             // u at offset 2, zero halo on each side
    __shared__ double s_g[NPAD];                 // M u^n + dt F of the current time step
    __shared__ double s_h[NPAD];                 // hfs: h_e (f(gp1) + f(gp2)) per element
    __shared__ double s_fdt[NPAD];               // dt F
    constexpr int CW = GAL ? 4 : 6;              // coefficients per row the projection reads: lo, di, up, R / the pentadiagonal form's six
    __shared__ __attribute__((aligned(16))) double s_coef[NPAD][CW];
#ifndef BG_ACC_BUDGET
#define BG_ACC_BUDGET (((256 - 2 * S * NB - 56) / 2) < 24 ? ((256 - 2 * S * NB - 56) / 2) : 24)
#endif
    // accumulators a pass may keep live: what 256 registers leave beside the 2 S NB fragment registers and ~56 others (measured at r = 40, N = 512: 16 ... 24 accumulators run alike, 30 spill into the MFMA loop and lose 25 %)
#ifndef BG_ACC_BUDGET_LSPG
    // LSPG in the pentadiagonal form (mfma_pass, PENTA): a pass over the column blocks cb0 .. cb1 keeps cb1 - cb0 + 1 operands
    // Z live next to its accumulators.  Measured at r = 40, N = 512 (B = 4096): 12: 2.16e7, 15 / 16: 2.17e7, 18 / 20: 2.20e7
    // (four passes: 15 + 13 + 17 + 20 accumulators, 56 B of scratch), 21: 2.13e7, 24: 2.10e7 (three passes, 136 B)
#define BG_ACC_BUDGET_LSPG (((256 - 2 * S * NB - 48) / 2) < 20 ? ((256 - 2 * S * NB - 48) / 2) : 20)
#endif
    constexpr int kAccBudget = PIV ? 64 : (GAL ? (BG_ACC_BUDGET) : (BG_ACC_BUDGET_LSPG));
#ifndef BG_NRED
#define BG_NRED 2
#endif
    constexpr int NRED = BG_NRED;
    __shared__ double s_red[NRED][RW][RW + 4];   // per wave pair {w, w + 2}:  Ar | [br, W^T u (Galerkin), 0, 0]
    __shared__ double s_wtu[4][RW];              // per-wave  W^T u (LSPG)
    __shared__ double s_edge[2][2][NB][4][4];    // the two Phi rows just below / above each WAVE's 16 S rows, per column block and t
    __shared__ double s_m[2][4][64];             // multipliers of the current / next panel
    __shared__ double s_diag[RW], s_y[RW];       // what the elimination leaves: diagonal and right-hand side
    __shared__ double s_q[RW];
    __shared__ double s_x[RW];                   // solution of the pivoted fallback
    __shared__ int s_bad[4];
    __shared__ int s_info;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction
    const int t = lane & 3, owner = 16 * w + (lane >> 2);
    const int N = a.N, r = a.r;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const int rowbase = owner * S;

    // ---- basis fragments: loaded once per workgroup, kept in registers for every sample ------------------------
    double frag[NB][S];                          // Phi[rowbase + s][4 c + t]
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const int col = 4 * c + t;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = rowbase + s;
            frag[c][s] = (i < N && col < r) ? a.Phi[(size_t)i * r + col] : 0.0;
        }
        if ((lane >> 2) == 0) {                  // the rows below this wave's first row (inside a wave: HaloLanes)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int il = rowbase - 1 - d;
                s_edge[0][d][c][w][t] = (il >= 0 && il < N && col < r) ? a.Phi[(size_t)il * r + col] : 0.0;
            }
        }
        if ((lane >> 2) == 15) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int ih = rowbase + S + d;
                s_edge[1][d][c][w][t] = (ih < N && col < r) ? a.Phi[(size_t)ih * r + col] : 0.0;
            }
        }
    }
    if (tid < 4) s_u[tid < 2 ? tid : NPAD + tid] = 0.0;      // halos [0], [1], [NPAD+2], [NPAD+3]

    // Issue priority (round 3).  Two workgroups share every SIMD, and the matrix instruction holds the issue port for 16
    // cycles (K2f: it overlaps with nothing): a wave in one of the dependent phases -- assembly, partial-system sums, the
    // cooperative elimination with its four serial pivots per panel, update, lift -- used to queue behind the other
    // workgroup's MFMA stream with every short instruction of its chain.  Those phases run at priority 3, the MFMA passes
    // at 0: the chain gets the port when it is ready, the MFMA stream fills the gaps.  LSPG 1.73e7 -> 1.83e7, Galerkin
    // 1.78e7 -> 1.87e7 (B = 4096, same box); raising it around the elimination alone gave half of that.
    __builtin_amdgcn_s_setprio(3);
    for (int slot = blockIdx.x; slot < a.B; slot += gridDim.x) {
        const int smp = a.order ? a.order[slot] : slot;
        if (PIV && !a.force_pivoted && a.info[smp] != BG_INFO_NEEDS_PIVOTING) continue;      // workgroup-uniform
        const double mu1 = a.mu1[smp], mu2 = a.mu2[smp];
        double* hist = a.hist + (size_t)smp * (size_t)(a.nsteps + 1) * (size_t)N;
        __syncthreads();                                       // previous sample's LDS reads are done
        // ---- per-sample constants (compute_forcing_vector :427-461, f_gp of :556-558) and the initial state ----
        for (int i = tid; i < NPAD; i += 256) {
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
        __syncthreads();

        long long cyc[6] = {0, 0, 0, 0, 0, 0};          // timing builds only: shader clocks per phase
        long long tick = kTiming ? (long long)__builtin_amdgcn_s_memtime() : 0;
        auto lap = [&](int i) {
            if constexpr (kTiming) {
                const long long now = (long long)__builtin_amdgcn_s_memtime();
                cyc[i] += now - tick;
                tick = now;
            }
        };
        const long long stamp0 = kTiming ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const long long real0 = kTiming ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
        int flags = 0, info_out = 0;
        bool aborted = false;                    // fast kernel: the multiplier guard tripped, the repair kernel redoes this sample
        for (int step = 0; step < a.nsteps && info_out == 0 && !aborted; ++step) {
            // ---- g = M u^n + dt F (`M @ U[:, n] + At*F`, :746); rows are revisited by the same thread below ------
            for (int i = tid; i < NPAD; i += 256) {
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
            int k = 0;
            bool more;
            lap(5);
            do {
                // Per-lane addresses are loop invariants of this whole kernel; the optimiser hoists them out of the time loop by
                // the dozen and then spills them.  An opaque copy of the thread index per iteration keeps them recomputed
                // (a few integer instructions) instead: scratch 160 -> 76 bytes (Galerkin), 280 -> 140 (LSPG).
                int tid_i = tid;
                asm volatile("" : "+v"(tid_i));
                const int lane = tid_i & 63, t = lane & 3, rowbase = (16 * w + (lane >> 2)) * S;
                const HaloLanes<NB> halo{s_edge, w, t, lane};
                // ---- assembly: A(u_k), R(u_k) per row into LDS ----------------------------------------------------
                if (!skip(16))
                for (int i = tid; i < NPAD; i += 256) {
                    double lo, di, up, R;
                    const bool in = i < N;
                    const MeshConst mc = make_mesh_const(h, a.dt, a.E, a.supg);   // wave-uniform; rebuilt here rather than held in 14 VGPRs
                    rom_assemble_row(i, N, s_u[i + 1], s_u[i + 2], (i + 1 < N) ? s_u[i + 3] : 0.0, in ? s_g[i] : 0.0,
                                     (in && i > 0) ? s_h[i - 1] : 0.0, (in && i < N - 1) ? s_h[i] : 0.0, mu1, mc,
                                     a.nonuniform, a.x, a.dt, a.E, lo, di, up, R);
                    s_coef[i][0] = lo; s_coef[i][1] = di; s_coef[i][2] = up; s_coef[i][3] = R;
                }
                __syncthreads();
                if constexpr (!GAL) {
                    // LSPG in the pentadiagonal form (mfma_pass, PENTA): row i of A^T A and (A^T R)_i from the rows i - 1, i, i + 1
                    // of A (J_k = [lo_k, di_k, up_k] at the columns k - 1, k, k + 1):
                    //   (A^T A)_{i,i-2} = up_{i-1} lo_{i-1}            (A^T A)_{i,i-1} = up_{i-1} di_{i-1} + di_i lo_i
                    //   (A^T A)_{i,i}   = up_{i-1}^2 + di_i^2 + lo_{i+1}^2
                    //   (A^T A)_{i,i+1} = di_i up_i + lo_{i+1} di_{i+1}    (A^T A)_{i,i+2} = lo_{i+1} up_{i+1}
                    //   (A^T R)_i       = up_{i-1} R_{i-1} + di_i R_i + lo_{i+1} R_{i+1}
                    // read by every thread for its own rows, a barrier, then written over the same rows.
                    double pen[NPAD / 256][6];
#pragma unroll
                    for (int ii = 0; ii < NPAD / 256; ++ii) {
                        const int i = tid + 256 * ii;
                        const bool hm = i > 0, hp = i + 1 < NPAD;
                        const double lm = hm ? s_coef[hm ? i - 1 : 0][0] : 0.0, dm = hm ? s_coef[hm ? i - 1 : 0][1] : 0.0;
                        const double um = hm ? s_coef[hm ? i - 1 : 0][2] : 0.0, Rm = hm ? s_coef[hm ? i - 1 : 0][3] : 0.0;
                        const double l0 = s_coef[i][0], d0 = s_coef[i][1], u0c = s_coef[i][2], R0 = s_coef[i][3];
                        const double lp = hp ? s_coef[hp ? i + 1 : 0][0] : 0.0, dp = hp ? s_coef[hp ? i + 1 : 0][1] : 0.0;
                        const double upp = hp ? s_coef[hp ? i + 1 : 0][2] : 0.0, Rp = hp ? s_coef[hp ? i + 1 : 0][3] : 0.0;
                        pen[ii][0] = um * lm;
                        pen[ii][1] = __builtin_fma(um, dm, d0 * l0);
                        pen[ii][2] = __builtin_fma(lp, lp, __builtin_fma(d0, d0, um * um));
                        pen[ii][3] = __builtin_fma(lp, dp, d0 * u0c);
                        pen[ii][4] = lp * upp;
                        pen[ii][5] = __builtin_fma(lp, Rp, __builtin_fma(d0, R0, um * Rm));
                    }
                    __syncthreads();
#pragma unroll
                    for (int ii = 0; ii < NPAD / 256; ++ii) {
                        const int i = tid + 256 * ii;
#pragma unroll
                        for (int e = 0; e < 6; ++e) s_coef[i][e] = pen[ii][e];
                    }
                    __syncthreads();
                }
                lap(0);
                // ---- projection on the matrix cores -----------------------------------------------------------------
                // as many passes over the rows as the accumulator budget of this instantiation demands (mfma_passes)
                __builtin_amdgcn_s_setprio(0);
                if constexpr (!skip(1))
                    mfma_passes<S, NB, true, RW, NRED, kAccBudget, 0, !GAL, CW>(frag, halo, s_coef, s_u, rowbase, t, w, lane, s_red, s_wtu);
                __builtin_amdgcn_s_setprio(3);
                __syncthreads();
                lap(1);
                // ---- reduced solve: load own columns (sum of the four waves' partials), eliminate ----------------
                auto entry = [&](int i, int j) -> double {               // (Ar | br | wtu)[i][j], j <= RW + 1
                    int rr = i, cc = j;
                    if (!GAL && j < RW && (i >> 2) > (j >> 2)) { rr = j; cc = i; }   // LSPG: mirror the lower blocks
                    return red_sum<NRED, RW>(s_red, rr, cc);
                };
                double xout = 0.0;
                if constexpr (PIV) {
                    if (tid == 0) s_info = 0;
                    __syncthreads();
                    if (w == 0) pivoted_solve<NB, GAL, NRED>(s_red, s_x, &s_info, lane, r);
                    __syncthreads();
                    xout = (lane < RW) ? s_x[lane] : 0.0;
                    if (s_info != 0 && info_out == 0) info_out = s_info;
                } else {
                    bool tripped;
                    xout = coop_gj_solve<NB, GAL, !skip(2), NRED>(s_red, s_m, s_diag, s_y, s_bad, w, lane, r, tripped);

                    if (!kTiming && tripped) aborted = true;
                }
                lap(2);
                // ---- q = Phi^T u_k + dq, err = |dq| / |q|  (:770-776) ---------------------------------------------
                double wtu = 0.0;
                if (lane < r) wtu = entry(lane, RW + 1);        // Phi^T u: the second column of the extra block (both forms)
                const double dq = (lane < r) ? xout : 0.0;
                const double qn = (lane < r) ? wtu + dq : 0.0;
                double nd, nq;
                wave_sum2(dq * dq, qn * qn, nd, nq);
                nd = sqrt(nd); nq = sqrt(nq);
                const double err = nd / nq;
                ++k;
                more = (err > a.tol) && (k < a.max_it) && info_out == 0 && !aborted;
                if (kTiming) more = k < 5;
                if (!(err - err == 0.0)) flags |= BG_FLAG_NONFINITE;
                if (k >= a.max_it) flags |= BG_FLAG_HIT_CAP;
                if (w == 0 && lane < RW) s_q[lane] = qn;
                __syncthreads();
                lap(3);
                // ---- lift u_{k+1} = Phi q from the register-resident basis (:773) --------------------------------
                if (!skip(8)) {
                    double qv[NB];
#pragma unroll
                    for (int c = 0; c < NB; ++c) qv[c] = s_q[4 * c + t];          // zero beyond r
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        double p = 0.0;
#pragma unroll
                        for (int c = 0; c < NB; ++c) p = __builtin_fma(frag[c][s], qv[c], p);
                        p += dpp_mov<0xB1>(p);             // quad_perm [1,0,3,2]: sum over the four t lanes
                        p += dpp_mov<0x4E>(p);             // quad_perm [2,3,0,1]
                        if (t == 0) {
                            const int i = rowbase + s;
                            s_u[i + 2] = (i < N) ? p : 0.0;
                        }
                    }
                }
                __syncthreads();
                lap(4);
            } while (more);
            // ---- U[:, n+1] = U1 (:779): one coalesced row ----------------------------------------------------------
            double* hrow = hist + (size_t)(step + 1) * N;
            for (int i = tid; i < N; i += 256) hrow[i] = s_u[i + 2];
            if (tid == 0) a.iters[(size_t)smp * a.nsteps + step] = k;
        }
        if (tid == 0) {
            a.flags[smp] = flags;
            a.info[smp] = aborted ? BG_INFO_NEEDS_PIVOTING : info_out;
            if (kTiming && a.nsteps >= 2) {      // diagnostic builds only: shader cycles and 100 MHz ticks of this sample, in kilo-units
                a.iters[(size_t)smp * a.nsteps] = (int)(((long long)__builtin_amdgcn_s_memtime() - stamp0) >> 10);
                a.iters[(size_t)smp * a.nsteps + 1] = (int)(((long long)__builtin_amdgcn_s_memrealtime() - real0) >> 10);
                if (a.nsteps >= 8)
                    for (int i = 0; i < 6; ++i) a.iters[(size_t)smp * a.nsteps + 2 + i] = (int)(cyc[i] >> 10);
            }
        }
    }
}

template <int S, int NB>
void launch_fused(int projection, int grid, hipStream_t st, const RomRunArgs& a)
{
    if (projection == BG_PROJ_GALERKIN)
        hipLaunchKernelGGL((rom_fused_kernel<S, NB, BG_PROJ_GALERKIN, false>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((rom_fused_kernel<S, NB, BG_PROJ_LSPG, false>), dim3(grid), dim3(256), 0, st, a);
}

// The repair kernel: samples the fast kernel gave up on (info = BG_INFO_NEEDS_PIVOTING), redone from u0 with the
// partial-pivoting solve.  One shape covers every N <= 512 and r <= 40 (zero padding); it is not a hot path.
void launch_repair(int projection, int grid, hipStream_t st, const RomRunArgs& a)
{
    if (projection == BG_PROJ_GALERKIN)
        hipLaunchKernelGGL((rom_fused_kernel<8, 10, BG_PROJ_GALERKIN, true>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((rom_fused_kernel<8, 10, BG_PROJ_LSPG, true>), dim3(grid), dim3(256), 0, st, a);
}

}  // namespace

extern "C" {

int bg_rom_run_max_r(void) { return 40; }

int bg_rom_run(int N, int B, int r, int nsteps, int projection, const double* x, const double* Phi, const double* u0,
               const double* mu1, const double* mu2, double dt, double E, double tol, int max_it, int options,
               double* hist, int32_t* iters, int32_t* flags, int32_t* info, const int32_t* order, void* stream)
{
    if (N < 2 || B < 0 || r < 1 || nsteps < 0 || max_it < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (projection != BG_PROJ_GALERKIN && projection != BG_PROJ_LSPG) return BG_ERR_PROJECTION;
    if (N > 512) return BG_ERR_UNSUPPORTED_N;
    if (r > 40) return BG_ERR_UNSUPPORTED_R;
    if (B == 0) return BG_OK;
    if (!x || !Phi || !u0 || !mu1 || !mu2 || !hist || !flags || !info || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    RomRunArgs a;
    a.x = x; a.Phi = Phi; a.u0 = u0; a.mu1 = mu1; a.mu2 = mu2; a.hist = hist; a.iters = iters; a.flags = flags;
    a.info = info; a.order = order; a.dt = dt; a.E = E; a.tol = tol; a.N = N; a.B = B; a.r = r; a.nsteps = nsteps; a.max_it = max_it;
    a.supg = options & BG_OPT_SUPG; a.nonuniform = (options & BG_OPT_NONUNIFORM) ? 1 : 0;
    a.force_pivoted = (options & BG_OPT_FORCE_PIVOTED) ? 1 : 0;
    const int cus = device_cu_count();
    const int grid = B < 2 * cus ? B : 2 * cus;              // two resident workgroups per CU
    const int grid_repair = B < cus ? B : cus;
    hipStream_t st = (hipStream_t)stream;
    const int nb = r <= 8 ? 2 : (r <= 24 ? 6 : 10);
    const int s4 = N <= 256 ? 4 : 8;
    if (!a.force_pivoted) {
        switch (s4 * 100 + nb) {
#ifndef BG_FUSED_ONLY_810                 // (experiments compile the headline instantiation alone)
            case 402: launch_fused<4, 2>(projection, grid, st, a); break;
            case 406: launch_fused<4, 6>(projection, grid, st, a); break;
            case 410: launch_fused<4, 10>(projection, grid, st, a); break;
            case 802: launch_fused<8, 2>(projection, grid, st, a); break;
            case 806: launch_fused<8, 6>(projection, grid, st, a); break;
#endif
            case 810: launch_fused<8, 10>(projection, grid, st, a); break;
            default: return BG_ERR_UNSUPPORTED_R;
        }
        const int rc = check_launch();
        if (rc != BG_OK) return rc;
    }
    if (kAblate < 0) launch_repair(projection, grid_repair, st, a);     // every workgroup leaves at once unless a sample is flagged
    return check_launch();
}

}  // extern "C"
