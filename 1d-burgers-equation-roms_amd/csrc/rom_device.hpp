// rom_device.hpp -- device pieces shared by the batched ROM kernels (rom.hip) and the fused ROM
// time-stepper (rom_fused.hip): one row of the Picard assembly, and the pivoted r x r solve of one
// wavefront.  reference: FEM/fem_burgers.py:730-776 (pod_prom_burgers inner body).
#pragma once
#include "fom_device.hpp"

namespace bg {

// Row i of A(u) and of R = A u - b for one sample (same closed forms and operation order as the FOM
// kernel).  um, u0, ur: u[i-1], u[i], u[i+1] (0 outside the mesh); gi = (M u^n + dt F)_i;
// hL, hR = hfs[i-1], hfs[i] (h_e (f(gp1) + f(gp2)) of the left / right element).  Rows i >= N are identity rows.
__device__ __forceinline__ void rom_assemble_row(int i, int N, double um, double u0, double ur, double gi, double hL,
                                                 double hR, double mu1, const MeshConst& mc, int nonuniform,
                                                 const double* __restrict__ x, double dt, double E, double& lo,
                                                 double& di, double& up, double& R)
{
    lo = 0.0; di = 1.0; up = 0.0;
    double rhs = 0.0;
    if (i < N) {
        double aoffL = mc.aoff, aoffR = mc.aoff, ddL = mc.dd1, ddR = mc.dd1;
        if (nonuniform && i > 0) {                          // per-element lengths
            const double hl = x[i] - x[i - 1];
            aoffL = hl / 6.0 - dt * E / hl; ddL = hl / 3.0 + dt * E / hl;
            if (i < N - 1) {
                const double hr = x[i + 1] - x[i];
                aoffR = hr / 6.0 - dt * E / hr; ddR = hr / 3.0 + dt * E / hr;
            }
        }
        if (i == 0) {
            rhs = mu1 - u0;                                   // Dirichlet row
        } else {
            const double wl = um + u0;                        // left element (i-1, i)
            lo = __builtin_fma(-mc.dt6, wl + u0, aoffL);
            const double tl = __builtin_fma(wl, u0 - um, -hL);
            const double sl = tl * rcp(fmax(fabs(wl), 2.0e-10));
            double b = __builtin_fma(-mc.kap, sl, gi);
            if (i < N - 1) {
                const double wr = u0 + ur;
                up = __builtin_fma(mc.dt6, wr + u0, aoffR);
                di = __builtin_fma(mc.dt6, um - ur, ddL + ddR);
                const double tr = __builtin_fma(wr, ur - u0, -hR);
                const double sr = tr * rcp(fmax(fabs(wr), 2.0e-10));
                b = __builtin_fma(mc.kap, sr, b);
            } else {
                di = __builtin_fma(mc.dt6, wl + u0, ddL);
            }
            rhs = __builtin_fma(-lo, um, b);
            rhs = __builtin_fma(-di, u0, rhs);
            rhs = __builtin_fma(-up, ur, rhs);
        }
    }
    R = -rhs;
}

__device__ __forceinline__ double readlane_f64(double v, int srclane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    // DPP butterfly on 32-bit keys; result valid in lane 63, then broadcast
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true); v = v > t ? v : t;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// x = solve(A, b) with partial pivoting, one wavefront per system: lane i holds row i of [A | b] in row[0..NMAX]
// (identity rows beyond n).  Rows are never moved: the pivot of step k is the not-yet-used lane with the largest
// |a[k]| (LAPACK gesv's choice up to ties in the top 32 bits), its row is broadcast with v_readlane.  Returns the
// solution component of lane k in lane k; info = 0 or (first zero-pivot step + 1).  reference: np.linalg.solve :767.
template <int NMAX>
__device__ __forceinline__ double lu_pivoted_wave(double (&row)[NMAX + 1], int lane, int& info)
{
    bool used = false;        // this lane's row already served as a pivot row
    int my_step = -1;         // elimination step at which it did
    info = 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        // pivot search on the high dword of |a[k]| (monotone for non-negative doubles)
        unsigned key = used ? 0u : (((unsigned)__double2hiint(row[k]) & 0x7fffffffu) + 1u);
        const unsigned best = wave_max_u32(key);
        const unsigned long long m = __ballot(key == best && !used);
        const int p = __builtin_ctzll(m);                       // lowest candidate lane
        const double piv = readlane_f64(row[k], p);
        if (piv == 0.0 && info == 0) info = k + 1;
        const double rp = rcp(piv);
        const bool is_p = lane == p;
        const double mult = (used || is_p) ? 0.0 : row[k] * rp;
#pragma unroll
        for (int j = k + 1; j <= NMAX; ++j) row[j] = __builtin_fma(-mult, readlane_f64(row[j], p), row[j]);
        if (is_p) { used = true; my_step = k; }
    }
    // back substitution, column oriented: the row that pivoted at step k holds U[k][*]
    double xout = 0.0;
#pragma unroll
    for (int k = NMAX - 1; k >= 0; --k) {
        const unsigned long long m = __ballot(my_step == k);
        const int p = __builtin_ctzll(m);
        const double xk = readlane_f64(row[NMAX], p) * rcp(readlane_f64(row[k], p));
        row[NMAX] = (my_step < k) ? __builtin_fma(-row[k], xk, row[NMAX]) : row[NMAX];
        xout = (lane == k) ? xk : xout;
    }
    return xout;
}

}  // namespace bg
