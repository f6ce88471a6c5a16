// rom_fused_device.hpp -- device pieces shared by the fused (one workgroup per sample) ROM kernels: the MFMA projection
// pass, the cooperative guarded elimination of the reduced system, the partial-pivoting solve of the repair kernels.
// Used by rom_fused.hip (POD: bg_rom_run) and rom_ann_fused.hip (POD-ANN: bg_ann_rom_run).
#pragma once
#include "rom_device.hpp"

// Phase ablation for tools/time_fused.py (never defined in the product build): BG_FUSED_ABLATE is a bit mask of phases to
// compile out (1 MFMA passes, 2 elimination, 8 lift, 16 assembly, 32 MFMA epilogue, 64 per-step operand formation); the
// iteration count is then fixed at 5 per time step, the multiplier guard is ignored and the repair kernel is not launched,
// so that the garbage values cannot change the control flow.  Such builds also stamp s_memtime / s_memrealtime per sample
// into iters[b][0..1] (the in-kernel clock, printed by tools/time_fused.py).
#ifndef BG_FUSED_ABLATE
#define BG_FUSED_ABLATE -1
#endif

namespace bg {
namespace fused {
constexpr int kAblate = BG_FUSED_ABLATE;
constexpr bool kTiming = kAblate >= 0;
constexpr bool skip(int bit) { return kTiming && (kAblate & bit) != 0; }


// LDS  *p += v  without a return value: ds_add_f64.  Callers give every address exactly one adder per barrier interval,
// so the sum does not depend on the order in which waves arrive.
__device__ __forceinline__ void lds_add_f64(double* p, double v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- one MFMA pass over this wave's rows: pairs (ca, cb) with ca in [CA0, CA1) (LSPG) / cb in [CA0, CA1) (Galerkin) ----
// Same operand layout and accumulation order as rom_reduce4_kernel (rom.hip).  frag[c][s] = Phi[rowbase + s][4 c + t]; the
// two halo rows Phi[rowbase - 1], Phi[rowbase + S] of every column block come from `halo(side, c)` when the first / last
// row step needs them (they are not kept in registers: 40 VGPRs fewer live across the whole kernel):
//   HaloTable   one private LDS slot per thread (s_halo[side][c][tid], conflict-free 8-byte reads): rom_ann_fused, whose
//               tangent changes every iteration and is written to LDS anyway;
//   HaloLanes   the neighbouring owner's fragment registers through ds_bpermute (lane -+ 4 holds the same column of the
//               row below / above), and a 2.5 KB LDS table for the four lanes per side whose neighbour is in another
//               wave: rom_fused, where the basis is constant -- the 40 KB table of the first form is what kept that kernel
//               at one workgroup per CU.
// LAST: this pass also carries the Phi^T u accumulators of the LSPG form / the [R, u] column of the Galerkin form.  The
// Galerkin passes split the B side (cb): a pass forms only the (A Phi) operands of its own column blocks, so two passes
// form every operand once (a split of the A side formed all of them twice: 240 instructions).
// NRED: how the four waves' partial systems meet.  4: one LDS buffer per wave, summed by the reader as
// (w0 + w1) + (w2 + w3) (the batched path's order).  2: waves 0, 1 store, a workgroup barrier, waves 2, 3 add theirs on
// top (every address receives exactly one add: the result does not depend on timing); the reader sums
// (w0 + w2) + (w1 + w3).  Half the LDS (28 KB instead of 56 KB at r = 40) for one more barrier per pass.
template <int NB>
struct HaloTable {
    const double (*tab)[NB][256];
    int tid;
    template <int S>
    __device__ __forceinline__ double operator()(int side, int c, const double (&)[NB][S], int = 0) const { return tab[side][c][tid]; }
};

template <int NB>
struct HaloLanes {
    const double (*edge)[2][NB][4][4];   // [side][depth][c][wave][t]: the two rows just outside each wave's 16 S rows
    int w, t, lane;
    // depth 0: the row next to this lane's block (rowbase - 1 / rowbase + S), depth 1: one further out (the pentadiagonal form)
    template <int S>
    __device__ __forceinline__ double operator()(int side, int c, const double (&frag)[NB][S], int depth = 0) const
    {
        const double mine = side == 0 ? frag[c][S - 1 - depth] : frag[c][depth];
        const int src = side == 0 ? lane - 4 : lane + 4;
        const double nb = from_lane_rot(mine, (src & 63) << 2);
        const bool outside = side == 0 ? lane < 4 : lane >= 60;
        return outside ? edge[side][depth][c][w][t] : nb;
    }
};

// PENTA (with GAL = true): the LSPG system in its pentadiagonal form, Ar = Phi^T (A^T A Phi), br = Phi^T (A^T R).  The caller
// hands in the five coefficients of row i of A^T A and w_i = (A^T R)_i (s_coef[i][0..5]); the pass is the Galerkin pass --
// A operands are the basis fragments themselves, every B operand Z = (A^T A) Phi is formed once -- restricted to the block
// pairs ca <= cb (the system is symmetric: the reader mirrors).  Against the (A Phi)^T (A Phi) form: 65 instead of 75 matrix
// instructions per row step, 10 operand formations instead of 24 per iteration, three passes instead of four at 24
// accumulators, and the register profile of the Galerkin kernel (the Y^T Y form kept nine fragment doubles in scratch).
template <int NB, bool GAL, bool PENTA>
constexpr int pass_rows(int c0, int c1)
{
    int n = 0;
    for (int c = c0; c < c1; ++c) n += PENTA ? c + 1 : (GAL ? NB : NB - c + 1);
    return n;
}

template <int S, int NB, bool GAL, int CA0, int CA1, bool LAST, int RW, int NRED = 4, bool PENTA = false, int CW = 4, class Halo>
__device__ __forceinline__ void mfma_pass(const double (&frag)[NB][S], const Halo& halo,
                                          const double (*__restrict__ s_coef)[CW], const double* __restrict__ s_u,
                                          int rowbase, int t, int w, int lane,
                                          double (*__restrict__ s_red)[RW][RW + 4], double (*__restrict__ s_wtu)[RW])
{
    static_assert(NRED == 4 || NRED == 2, "four per-wave partial systems, or two shared by wave pairs");
    static_assert(!PENTA || (GAL && CW >= 6), "the pentadiagonal form runs the Galerkin pass on six coefficients per row");
    constexpr int NROW = pass_rows<NB, GAL, PENTA>(CA0, CA1);
    constexpr int NACC = NROW + (LAST ? NB : 0);
    double acc[NACC];
#pragma unroll
    for (int p = 0; p < NACC; ++p) acc[p] = 0.0;
    // B operands of row step s: Y = (A Phi) rows, X = extra block [R, u, 0, 0]
    auto form_y = [&](int s, int c, double lo, double di, double up) -> double {
        const double below = (s == 0) ? halo(0, c, frag) : frag[c][s == 0 ? 0 : s - 1];
        const double above = (s == S - 1) ? halo(1, c, frag) : frag[c][s == S - 1 ? s : s + 1];
        double y = lo * below;
        y = __builtin_fma(di, frag[c][s], y);
        return __builtin_fma(up, above, y);
    };
    // No software pipelining here: measured (tools/mfma_valu_bench.hip) the fp64 4x4x4 MFMA does not overlap with vector
    // ALU work of the same wave -- 110 MFMAs take 882 ns alone and 882 + 2.25 ns per interleaved v_fma_f64 -- so the
    // phase costs the MFMAs plus every other instruction; interleaving only adds hazard s_nops (498 against 103).
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ss = skip(64) ? 0 : s;         // (64: timing only)
        const int i = rowbase + ss;
        const double lo = s_coef[i][0], di = s_coef[i][1], up = s_coef[i][2], R = s_coef[i][3];
        const double ui = s_u[i + 2];
        double X;
        if constexpr (PENTA) X = (t == 0) ? s_coef[i][5] : ((t == 1) ? ui : 0.0);      // [A^T R, u, 0, 0]
        else X = (t == 0) ? R : ((t == 1) ? ui : 0.0);
        if constexpr (GAL) {
            double Y[CA1 - CA0];
            if constexpr (PENTA) {
                // Z = (A^T A Phi) row: coefficients of the rows i - 2 .. i + 2 in s_coef[i][0..4] (lo, di, up, R above hold 0..3)
                const double p2 = s_coef[i][4];
#pragma unroll
                for (int c = CA0; c < CA1; ++c) {
                    const double fm2 = (s >= 2) ? frag[c][s >= 2 ? s - 2 : 0] : halo(0, c, frag, 1 - s);
                    const double fm1 = (s >= 1) ? frag[c][s >= 1 ? s - 1 : 0] : halo(0, c, frag, 0);
                    const double fp1 = (s <= S - 2) ? frag[c][s <= S - 2 ? s + 1 : 0] : halo(1, c, frag, 0);
                    const double fp2 = (s <= S - 3) ? frag[c][s <= S - 3 ? s + 2 : 0] : halo(1, c, frag, s - (S - 2));
                    double z = lo * fm2;
                    z = __builtin_fma(di, fm1, z);
                    z = __builtin_fma(up, frag[c][s], z);
                    z = __builtin_fma(R, fp1, z);
                    Y[c - CA0] = __builtin_fma(p2, fp2, z);
                }
            } else {
#pragma unroll
                for (int c = CA0; c < CA1; ++c) Y[c - CA0] = form_y(ss, c, lo, di, up);
            }
            int p = 0;
#pragma unroll
            for (int ca = 0; ca < NB; ++ca) {
#pragma unroll
                for (int cb = CA0; cb < CA1; ++cb) {
                    if (PENTA && ca > cb) continue;
                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(frag[ca][s], Y[cb - CA0], acc[p], 0, 0, 0);
                    ++p;
                }
            }
            if constexpr (LAST) {
#pragma unroll
                for (int ca = 0; ca < NB; ++ca, ++p)
                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(frag[ca][s], X, acc[p], 0, 0, 0);
            }
        } else {
            // LSPG, pairs (ca in the pass's range, cb >= ca).  Only the range's own operands stay live; the columns beyond it
            // are formed one at a time, used by every ca of the range and dropped (all NB - CA0 of them live at once spilled
            // into the matrix loop: 2.8 GB of scratch re-fetches per 40-step launch against 0.07 GB for the Galerkin form)
            constexpr int NA = CA1 - CA0;
            auto pidx = [](int ca, int cb) {         // as for_each_acc enumerates them: ca major, cb = ca .. NB
                int p = 0;
                for (int c = CA0; c < ca; ++c) p += NB - c + 1;
                return p + (cb - ca);
            };
            double Ya[NA];
#pragma unroll
            for (int ca = CA0; ca < CA1; ++ca) Ya[ca - CA0] = form_y(ss, ca, lo, di, up);
#pragma unroll
            for (int ca = CA0; ca < CA1; ++ca) {
#pragma unroll
                for (int cb = ca; cb < CA1; ++cb)
                    acc[pidx(ca, cb)] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ya[ca - CA0], Ya[cb - CA0], acc[pidx(ca, cb)], 0, 0, 0);
            }
#pragma unroll
            for (int cb = CA1; cb < NB; ++cb) {
                const double Yb = form_y(ss, cb, lo, di, up);
#pragma unroll
                for (int ca = CA0; ca < CA1; ++ca)
                    acc[pidx(ca, cb)] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ya[ca - CA0], Yb, acc[pidx(ca, cb)], 0, 0, 0);
            }
#pragma unroll
            for (int ca = CA0; ca < CA1; ++ca)
                acc[pidx(ca, NB)] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ya[ca - CA0], X, acc[pidx(ca, NB)], 0, 0, 0);
            if constexpr (LAST) {
#pragma unroll
                for (int ca = 0; ca < NB; ++ca)
                    acc[NROW + ca] = __builtin_amdgcn_mfma_f64_4x4x4f64(frag[ca][s], X, acc[NROW + ca], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // nothing moves across a step: bounded register pressure
    }
    // sum the four block partials of every pair (lanes differing in bits 2..3), park them per wave (pair) in LDS
    const int oi = lane >> 4, oj = lane & 3;
    const bool writer = ((lane >> 2) & 3) == 3;
    if constexpr (skip(32)) {                    // timing only: keep the MFMAs alive, drop the block sums and stores
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NACC; ++q) v += acc[q];
        if (writer) s_red[NRED == 4 ? w : (w & 1)][oi][oj] = v;
        if constexpr (NRED == 2) __syncthreads();
        return;
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        double v = acc[q];
        v += dpp_mov<0x114>(v);                  // row_shr:4
        v += dpp_mov<0x118>(v);                  // row_shr:8 -> lanes with blk == 3 hold the sum
        acc[q] = v;
    }
    // visit every accumulator with its place in the reduced system: f(p, row block, column block)
    auto for_each_acc = [&](auto&& f) {
        int p = 0;
        if constexpr (GAL) {
#pragma unroll
            for (int ca = 0; ca < NB; ++ca) {
#pragma unroll
                for (int cb = CA0; cb < CA1; ++cb) {
                    if (PENTA && ca > cb) continue;
                    f(p, ca, cb);
                    ++p;
                }
            }
            if constexpr (LAST) {
#pragma unroll
                for (int ca = 0; ca < NB; ++ca, ++p) f(p, ca, NB);
            }
        } else {
#pragma unroll
            for (int ca = CA0; ca < CA1; ++ca) {
#pragma unroll
                for (int cb = ca; cb <= NB; ++cb, ++p) f(p, ca, cb);
            }
        }
    };
    if constexpr (NRED == 4) {
        for_each_acc([&](int p, int ca, int cb) { if (writer) s_red[w][4 * ca + oi][4 * cb + oj] = acc[p]; });
    } else {
        double (*dst)[RW + 4] = s_red[w & 1];
        if (w < 2) for_each_acc([&](int p, int ca, int cb) { if (writer) dst[4 * ca + oi][4 * cb + oj] = acc[p]; });
        __syncthreads();
        if (w >= 2) for_each_acc([&](int p, int ca, int cb) { if (writer) lds_add_f64(&dst[4 * ca + oi][4 * cb + oj], acc[p]); });
    }
    if constexpr (!GAL && LAST) {
        constexpr int P0 = NROW;
#pragma unroll
        for (int ca = 0; ca < NB; ++ca)
            if (writer && oj == 1) s_wtu[w][4 * ca + oi] = acc[P0 + ca];
    }
}

// ---- the projection as a sequence of passes sized for a register budget -------------------------------------------
// A pass keeps NACC accumulators live next to the basis fragments.  With two workgroups per CU a lane has 256 registers,
// 2 S NB of them hold the fragments, so the (NB + 1) NB (Galerkin) / NB (NB + 1) / 2 + 2 NB (LSPG) accumulators are
// produced in as many passes as a budget of BUDGET accumulators demands.  Galerkin passes split the column blocks cb of
// the B side, LSPG passes the row blocks ca; the last pass carries the [R, u] / Phi^T u extras.
template <int NB, bool GAL, bool PENTA = false>
constexpr int pass_acc_count(int c0, int c1)
{
    return pass_rows<NB, GAL, PENTA>(c0, c1) + (c1 == NB ? NB : 0);
}

template <int NB, bool GAL, int BUDGET, bool PENTA = false>
constexpr int pass_end(int c0)
{
    int c1 = c0 + 1;
    while (c1 < NB && pass_acc_count<NB, GAL, PENTA>(c0, c1 + 1) <= BUDGET) ++c1;
    return c1;
}

template <int S, int NB, bool GAL, int RW, int NRED, int BUDGET, int C0 = 0, bool PENTA = false, int CW = 4, class Halo>
__device__ __forceinline__ void mfma_passes(const double (&frag)[NB][S], const Halo& halo,
                                            const double (*__restrict__ s_coef)[CW], const double* __restrict__ s_u,
                                            int rowbase, int t, int w, int lane,
                                            double (*__restrict__ s_red)[RW][RW + 4], double (*__restrict__ s_wtu)[RW])
{
    constexpr int C1 = pass_end<NB, GAL, BUDGET, PENTA>(C0);
    mfma_pass<S, NB, GAL, C0, C1, C1 == NB, RW, NRED, PENTA, CW>(frag, halo, s_coef, s_u, rowbase, t, w, lane, s_red, s_wtu);
    if constexpr (C1 < NB)
        mfma_passes<S, NB, GAL, RW, NRED, BUDGET, C1, PENTA, CW>(frag, halo, s_coef, s_u, rowbase, t, w, lane, s_red, s_wtu);
}

// ---- cooperative unpivoted elimination (see the file header) ----------------------------------------------------
template <int NB>
struct LuRegs {
    static constexpr int NSLOT = (NB + 3) / 4;
    double col[NSLOT][4];     // slot s = column block w + 4 s
    double rhs;               // wave 3 only
};

template <int NB>
__device__ __forceinline__ void lu_apply_block(double (&c)[4], const double (&m)[4], int p)
{
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {             // step-major: the eight readlanes of a step, then its four FMAs
        double piv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) piv[t] = readlane_f64(c[t], 4 * p + kk);
#pragma unroll
        for (int t = 0; t < 4; ++t) c[t] = __builtin_fma(-m[kk], piv[t], c[t]);
    }
}

// factor the panel held in c[0..3] (columns 4p .. 4p+3, pivots in lanes 4p .. 4p+3); multipliers -> sm[kk][lane]
// gmax: running maximum of the sub-diagonal |multipliers| (the partial-pivoting guard); zero_piv: a pivot was exactly 0
__device__ __forceinline__ void lu_factor_panel(double (&c)[4], int p, int lane, double (*__restrict__ sm)[64], double& gmax,
                                                bool& zero_piv)
{
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * p + kk;
        const double piv = readlane_f64(c[kk], k);
        const double rp = rcp(piv);
        const double m = (lane != k) ? c[kk] * rp : 0.0;          // rows ABOVE the pivot too (Gauss-Jordan, see the kernel)
        gmax = fmax(gmax, (lane > k) ? fabs(m) : 0.0);            // branch-free: a short-circuit here costs 3.7 us per solve
        zero_piv = zero_piv | (piv == 0.0);
#pragma unroll
        for (int jj = kk + 1; jj < 4; ++jj) c[jj] = __builtin_fma(-m, readlane_f64(c[jj], k), c[jj]);
        sm[kk][lane] = m;
    }
}

// The partial-pivoting solve of the repair kernel (PIV): one wave reloads the summed system from the per-wave partials and
// runs the routine of bg_lu_solve.  It lives in a kernel of its own: as a cold branch (even out of line) inside the fast
// kernel it cost 4.7 us per iteration through the register allocation around the call site.
// sum of the per-wave (NRED = 4) / per-wave-pair (NRED = 2, see mfma_pass) partial systems
template <int NRED, int RW>
__device__ __forceinline__ double red_sum(const double (*__restrict__ s_red)[RW][RW + 4], int rr, int cc)
{
    if constexpr (NRED == 4) return (s_red[0][rr][cc] + s_red[1][rr][cc]) + (s_red[2][rr][cc] + s_red[3][rr][cc]);
    else return s_red[0][rr][cc] + s_red[1][rr][cc];
}

template <int NB, bool GAL, int NRED = 4>
__device__ __forceinline__ void pivoted_solve(const double (*__restrict__ s_red)[4 * NB][4 * NB + 4],
                                              double* __restrict__ s_x, int* __restrict__ s_info, int lane, int r)
{
    constexpr int RW = 4 * NB;
    auto entry = [&](int i, int j) -> double {
        int rr = i, cc = j;
        if (!GAL && j < RW && (i >> 2) > (j >> 2)) { rr = j; cc = i; }
        return red_sum<NRED, RW>(s_red, rr, cc);
    };
    double row[RW + 1];
#pragma unroll
    for (int j = 0; j < RW; ++j) row[j] = (lane < r && j < r) ? entry(lane, j) : ((lane == j) ? 1.0 : 0.0);
    row[RW] = (lane < r) ? -entry(lane, RW) : 0.0;
    int info;
    const double xs = lu_pivoted_wave<RW>(row, lane, info);
    if (lane < RW) s_x[lane] = xs;
    if (lane == 0) *s_info = info;
}

// ---- the reduced solve of one iteration by all four waves (see rom_fused.hip's header) ---------------------------
// In: the four per-wave partial systems s_red (+ mirror for LSPG); r = live unknowns (the rest is identity padding).
// Out: x_k in lane k of EVERY wave (identical values); tripped = a multiplier above 1 or a zero pivot was seen (then x is
// not to be used).  Contains workgroup barriers: all 256 threads call it.  ELIM = false compiles the panels out (timing).
template <int NB, bool GAL, bool ELIM = true, int NRED = 4>
__device__ __forceinline__ double coop_gj_solve(const double (*__restrict__ s_red)[4 * NB][4 * NB + 4],
                                                double (*__restrict__ s_m)[4][64], double* __restrict__ s_diag,
                                                double* __restrict__ s_y, int* __restrict__ s_bad, int w, int lane, int r,
                                                bool& tripped)
{
    constexpr int RW = 4 * NB;
    constexpr int NSLOT = LuRegs<NB>::NSLOT;
    // `lane` is made opaque here: the per-lane LDS addresses and mirror selects of the load below are loop invariants, and
    // left to itself the optimiser hoists them out of the time loop and keeps (or spills) two dozen registers for them
    asm volatile("" : "+v"(lane));
    auto entry = [&](int i, int j) -> double {               // (Ar | br)[i][j], j <= RW
        int rr = i, cc = j;
        if (!GAL && j < RW && (i >> 2) > (j >> 2)) { rr = j; cc = i; }   // LSPG: mirror the lower blocks
        return red_sum<NRED, RW>(s_red, rr, cc);
    };
    LuRegs<NB> lu;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int b = w + 4 * s;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int j = 4 * b + tt;
            double v = 0.0;
            if (b < NB && lane < RW) v = (lane >= r || j >= r) ? ((lane == j) ? 1.0 : 0.0) : entry(lane, j);
            lu.col[s][tt] = v;
        }
    }
    lu.rhs = (w == 3 && lane < r) ? -entry(lane, RW) : 0.0;           // solve(Ar, -br)
    double gmax = 0.0;
    bool zero_piv = false;
    if (w == 0) lu_factor_panel(lu.col[0], 0, lane, s_m[0], gmax, zero_piv);
    __syncthreads();
    double mprev[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < (ELIM ? NB : 0); ++p) {
        double m[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) m[kk] = s_m[p & 1][kk][lane];
        const int nxt = p + 1;
        const bool owner_next = nxt < NB && w == (nxt & 3);
        const bool owner_this = p > 0 && w == (p & 3);       // deferred panel p - 1 for its later blocks
        if (owner_next) {                                     // look-ahead: next panel first ...
            lu_apply_block<NB>(lu.col[nxt >> 2], m, p);
            lu_factor_panel(lu.col[nxt >> 2], nxt, lane, s_m[nxt & 1], gmax, zero_piv);
        }
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int b = w + 4 * s;
            if (b > p && b < NB && b != nxt && !owner_next) {   // ... and its other blocks one panel later,
                if (owner_this) lu_apply_block<NB>(lu.col[s], mprev, p - 1);   // so that no wave carries
                lu_apply_block<NB>(lu.col[s], m, p);                           // factor + three updates
            }
        }
        if (w == 3) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) lu.rhs = __builtin_fma(-m[kk], readlane_f64(lu.rhs, 4 * p + kk), lu.rhs);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) mprev[kk] = m[kk];
        if (p + 1 < NB) __syncthreads();
    }
    // The multipliers cover the rows above the pivot as well (the FMAs run on all 64 lanes anyway), so what is
    // left is diagonal: x_k = y_k / d_k, no back substitution.  Publish d (owner of each column) and y (wave 3).
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int b = w + 4 * s;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            if (b < NB && lane == 4 * b + tt) s_diag[lane] = lu.col[s][tt];
    }
    if (w == 3 && lane < RW) s_y[lane] = lu.rhs;
    {
        const unsigned long long anybad = __ballot(zero_piv | !(gmax <= 1.0));
        if (lane == 0) s_bad[w] = anybad != 0ull;
    }
    __syncthreads();
    tripped = (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) != 0;                  // workgroup-uniform
    return (lane < RW) ? s_y[lane] * rcp(s_diag[lane]) : 0.0;
}

}  // namespace fused
}  // namespace bg
