// fom_device.hpp -- per-wavefront building blocks of the Burgers FOM Picard iteration.
//
// Row distribution: lane p of the wave owns the R consecutive mesh rows
// [p*R, p*R+R), so N <= 64*R rows fit one wavefront; rows >= N are identity rows.
// All per-row state lives in registers (fully unrolled, compile-time R).
//
// Arithmetic restated from the reference (FEM/fem_burgers.py), closed form for
// 2-node elements on a uniform mesh:
//   convection  :389-425   c1 = (2ul+ur)/6, c2 = (ul+2ur)/6
//   forcing     :427-461   2-pt Gauss of 0.02*exp(mu2 x)
//   SUPG        :500-581   s_e = tau*(ubar*du - (f1+f2)/2), tau = 0.5h/(2 max(|ubar|,1e-10))
//   system      :676-689   A = M + dt C + dt E K, row 0 <- e0, b = M u^n + dt F - dt S, b0 = mu1
#pragma once
#include "wave_ops.hpp"

// reciprocal used inside the Picard iteration (see wave_ops.hpp: rcp = 2 Newton steps, rcp1 = 1)
#ifndef BG_RCP
#define BG_RCP rcp1
#endif

namespace bg {

constexpr double GP_A = 0.78867513459481287;  // (1 + 1/sqrt(3)) / 2
constexpr double GP_B = 0.21132486540518713;  // (1 - 1/sqrt(3)) / 2

struct MeshConst {
    double h;      // uniform element length
    double aoff;   // h/6 - dt*E/h     (off-diagonal base)
    double dd2;    // 2*(h/3 + dt*E/h) (interior diagonal base)
    double dd1;    // h/3 + dt*E/h     (last-row diagonal base)
    double dt6;    // dt/6
    double kap;    // 0.25*dt  (0 when SUPG is off)
    double h6;     // h/6
};

__device__ __forceinline__ MeshConst make_mesh_const(double h, double dt, double E, int supg)
{
    MeshConst c;
    c.h = h;
    double eh = dt * E / h;
    c.aoff = h / 6.0 - eh;
    c.dd1 = h / 3.0 + eh;
    c.dd2 = 2.0 * c.dd1;
    c.dt6 = dt / 6.0;
    c.kap = supg ? 0.25 * dt : 0.0;
    c.h6 = h / 6.0;
    return c;
}

// Per-sample constants: hfs[j] = h*(f(gp1)+f(gp2)) of the element to the right of local
// row j, and fdt[j] = dt*F_i of row i = row0+j.
template <int R>
__device__ __forceinline__ void forcing_setup(const double* __restrict__ x, int N, int row0, double mu2,
                                              double h, double dt, double (&hfs)[R], double (&fdt)[R])
{
    double frPrev = 0.0;   // right-node load of element row0-1
    {
        int e = row0 - 1;
        if (e >= 0 && e < N - 1) {
            double xl = x[e], xr = x[e + 1];
            double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
            double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
            frPrev = (f1 * GP_B + f2 * GP_A) * (0.5 * h);
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int e = row0 + j;
        double fl = 0.0, fr = 0.0, fs = 0.0;
        if (e < N - 1) {
            double xl = x[e], xr = x[e + 1];
            double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
            double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
            fl = (f1 * GP_A + f2 * GP_B) * (0.5 * h);
            fr = (f1 * GP_B + f2 * GP_A) * (0.5 * h);
            fs = f1 + f2;
        }
        hfs[j] = h * fs;
        fdt[j] = (e < N) ? dt * (frPrev + fl) : 0.0;
        frPrev = fr;
    }
}

// g = M u^n + dt F  (constant over the Picard iterations of one time step).  uL / uR: u of the row below
// this lane's first row / above its last row (0 outside the mesh).
// gstore(j, value): where row j of g goes (registers, or LDS when a kernel parks it there)
template <int R, bool FULL, typename GStore>
__device__ __forceinline__ void mass_rhs_core_to(const MeshConst& c, int N, int row0, const double (&u)[R], double uL,
                                                 double uR, const double (&fdt)[R], GStore&& gstore)
{
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double um = (j == 0) ? uL : u[j - 1];
        const double up = (j == R - 1) ? uR : u[j + 1];
        const int i = row0 + j;
        double inner = __builtin_fma(4.0, u[j], um) + up;
        double last = __builtin_fma(2.0, u[j], um);
        double v = __builtin_fma(c.h6, (i == N - 1) ? last : inner, fdt[j]);
        gstore(j, (!FULL && i >= N) ? 0.0 : v);
    }
}

template <int R, bool FULL>
__device__ __forceinline__ void mass_rhs_core(const MeshConst& c, int N, int row0, const double (&u)[R], double uL,
                                              double uR, const double (&fdt)[R], double (&g)[R])
{
    mass_rhs_core_to<R, FULL>(c, N, row0, u, uL, uR, fdt, [&](int j, double v) { g[j] = v; });
}

template <int R, bool FULL>
__device__ __forceinline__ void mass_rhs(const MeshConst& c, int N, int row0, const double (&u)[R],
                                         const double (&fdt)[R], double (&g)[R])
{
    mass_rhs_core<R, FULL>(c, N, row0, u, from_lane_below(u[R - 1]), from_lane_above(u[0]), fdt, g);
}

// One assembly: diagonals lo/di/up of A(u) and rhs = b - A u  (= -R of the reference), in two halves so that
// the halo values can come from DPP (one wave per sample) or LDS (one workgroup per sample).
// p1: off-diagonals and the SUPG element terms se[];  p2: diagonal, right-hand side, special rows.
// `first` / `last_lane` mark the lanes that own global row 0 / (FULL only) the last row.
// hf(j): hfs of local row j (a register array, or an LDS read when a kernel parks it there)
template <int R, typename HF>
__device__ __forceinline__ void assemble_p1_from(const MeshConst& c, const double (&u)[R], double uL, double uR,
                                                 HF&& hf, double (&lo)[R], double (&up)[R], double (&se)[R])
{
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double ur = (j == R - 1) ? uR : u[j + 1];
        const double w = u[j] + ur;
        const double dif = ur - u[j];
        up[j] = __builtin_fma(c.dt6, w + u[j], c.aoff);
        if (j + 1 < R) lo[j + 1] = __builtin_fma(-c.dt6, w + ur, c.aoff);
        const double mx = fmax(fabs(w), 2.0e-10);
        const double t = __builtin_fma(w, dif, -hf(j));
        se[j] = t * BG_RCP(mx);
    }
    lo[0] = __builtin_fma(-c.dt6, __builtin_fma(2.0, u[0], uL), c.aoff);
}

template <int R>
__device__ __forceinline__ void assemble_p1(const MeshConst& c, const double (&u)[R], double uL, double uR,
                                            const double (&hfs)[R], double (&lo)[R], double (&up)[R],
                                            double (&se)[R])
{
    assemble_p1_from<R>(c, u, uL, uR, [&](int j) { return hfs[j]; }, lo, up, se);
}

template <int R, bool FULL, typename GF>
__device__ __forceinline__ void assemble_p2_from(const MeshConst& c, int N, int row0, bool first, bool last_lane,
                                                 double mu1, const double (&u)[R], double uL, double uR, double seL,
                                                 GF&& g, const double (&se)[R], double (&lo)[R],
                                                 double (&di)[R], double (&up)[R], double (&rhs)[R])
{
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double um = (j == 0) ? uL : u[j - 1];
        const double ur = (j == R - 1) ? uR : u[j + 1];
        const double sm = (j == 0) ? seL : se[j - 1];
        const int i = row0 + j;
        double d = __builtin_fma(c.dt6, um - ur, c.dd2);
        double b = __builtin_fma(-c.kap, sm, g(j));
        double bb = __builtin_fma(c.kap, se[j], b);
        double l = lo[j], p = up[j];
        // special rows: last real row has no right element; rows >= N are identity
        const bool is_last = FULL ? (j == R - 1 && last_lane) : (i == N - 1);
        if (FULL ? (j == R - 1) : true) {
            double dl = __builtin_fma(c.dt6, __builtin_fma(2.0, u[j], um), c.dd1);
            d = is_last ? dl : d;
            p = is_last ? 0.0 : p;
            bb = is_last ? b : bb;
        }
        if (!FULL) {
            const bool pad = i >= N;
            d = pad ? 1.0 : d;
            p = pad ? 0.0 : p;
            l = pad ? 0.0 : l;
            bb = pad ? 0.0 : bb;   // u is 0 on padded rows, so rhs becomes 0
        }
        if (j == 0) {           // Dirichlet row (global row 0)
            d = first ? 1.0 : d;
            p = first ? 0.0 : p;
            l = first ? 0.0 : l;
            bb = first ? mu1 : bb;
        }
        double r = __builtin_fma(-l, um, bb);
        r = __builtin_fma(-d, u[j], r);
        r = __builtin_fma(-p, ur, r);
        lo[j] = l; di[j] = d; up[j] = p; rhs[j] = r;
    }
}

template <int R, bool FULL>
__device__ __forceinline__ void assemble_p2(const MeshConst& c, int N, int row0, bool first, bool last_lane,
                                            double mu1, const double (&u)[R], double uL, double uR, double seL,
                                            const double (&g)[R], const double (&se)[R], double (&lo)[R],
                                            double (&di)[R], double (&up)[R], double (&rhs)[R])
{
    assemble_p2_from<R, FULL>(c, N, row0, first, last_lane, mu1, u, uL, uR, seL, [&](int j) { return g[j]; }, se, lo, di,
                              up, rhs);
}

template <int R, bool FULL>
__device__ __forceinline__ void assemble(const MeshConst& c, int N, int row0, double mu1,
                                         const double (&u)[R], const double (&g)[R],
                                         const double (&hfs)[R], double (&lo)[R], double (&di)[R],
                                         double (&up)[R], double (&rhs)[R])
{
    const int lane = lane_id();
    const double uL = from_lane_below(u[R - 1]);
    const double uR = from_lane_above(u[0]);
    double se[R];
    assemble_p1<R>(c, u, uL, uR, hfs, lo, up, se);
    const double seL = from_lane_below(se[R - 1]);
    assemble_p2<R, FULL>(c, N, row0, lane == 0, lane == 63, mu1, u, uL, uR, seL, g, se, lo, di, up, rhs);
}

// ------------------------------------------------------------------------------------
// General (non-uniform) mesh variants.  Element e = (i, i+1) has length h_e = x[i+1]-x[i];
// the lane keeps h_e/6 and dt*E/h_e of its R right-hand elements plus the one left of its
// first row (index 0).  Same closed forms as above with per-element constants
// (SURVEY.md Appendix A); rows >= N are identity rows.  Slower than the uniform path
// (more live registers); used only when the host reports a non-uniform mesh.
// ------------------------------------------------------------------------------------
template <int R>
struct ElemGeom {
    double h6[R + 1];   // h_e / 6        (index j+1: element right of local row j; index 0: left of row 0)
    double eh[R + 1];   // dt * E / h_e
};

template <int R>
__device__ __forceinline__ void geom_setup(const double* __restrict__ x, int N, int row0, double dt, double E,
                                           ElemGeom<R>& gm)
{
#pragma unroll
    for (int j = 0; j <= R; ++j) {
        const int e = row0 + j - 1;                       // element (e, e+1)
        double h = 1.0;
        if (e >= 0 && e < N - 1) h = x[e + 1] - x[e];
        gm.h6[j] = (e >= 0 && e < N - 1) ? h / 6.0 : 0.0;
        gm.eh[j] = (e >= 0 && e < N - 1) ? dt * E / h : 0.0;
    }
}

template <int R>
__device__ __forceinline__ void forcing_setup_general(const double* __restrict__ x, int N, int row0, double mu2,
                                                      double dt, double (&hfs)[R], double (&fdt)[R])
{
    double frPrev = 0.0;
#pragma unroll
    for (int j = -1; j < R; ++j) {
        const int e = row0 + j;
        double fl = 0.0, fr = 0.0, hf = 0.0;
        if (e >= 0 && e < N - 1) {
            const double xl = x[e], xr = x[e + 1], h = xr - xl;
            const double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
            const double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
            fl = (f1 * GP_A + f2 * GP_B) * (0.5 * h);
            fr = (f1 * GP_B + f2 * GP_A) * (0.5 * h);
            hf = h * (f1 + f2);
        }
        if (j >= 0) {
            hfs[j] = hf;
            fdt[j] = (e < N) ? dt * (frPrev + fl) : 0.0;
        }
        frPrev = fr;
    }
}

// core with the halo values passed in (uL = u of the row below this lane's first, uR = u of the row above its last)
template <int R>
__device__ __forceinline__ void mass_rhs_general_core(const ElemGeom<R>& gm, int N, int row0, const double (&u)[R],
                                                      double uL, double uR, const double (&fdt)[R], double (&g)[R])
{
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double um = (j == 0) ? uL : u[j - 1];
        const double up = (j == R - 1) ? uR : u[j + 1];
        const int i = row0 + j;
        // (M u)_i = h_{i-1}/6 (u_{i-1} + 2 u_i) + h_i/6 (2 u_i + u_{i+1}); h6 is 0 outside the mesh
        double v = gm.h6[j] * __builtin_fma(2.0, u[j], um);
        v = __builtin_fma(gm.h6[j + 1], __builtin_fma(2.0, u[j], up), v);
        g[j] = (i >= N) ? 0.0 : v + fdt[j];
    }
}

template <int R>
__device__ __forceinline__ void mass_rhs_general(const ElemGeom<R>& gm, int N, int row0, const double (&u)[R],
                                                 const double (&fdt)[R], double (&g)[R])
{
    mass_rhs_general_core<R>(gm, N, row0, u, from_lane_below(u[R - 1]), from_lane_above(u[0]), fdt, g);
}

// assemble_general in two halves so that the halo values can come from anywhere (DPP inside one wave,
// LDS across the waves of a workgroup).  p1: off-diagonals and the SUPG element terms se[];
// p2: diagonal, right-hand side, special rows.  `first` marks the lane that owns global row 0.
template <int R>
__device__ __forceinline__ void assemble_general_p1(const ElemGeom<R>& gm, double dt, const double (&u)[R], double uL,
                                                    double uR, const double (&hfs)[R], double (&lo)[R],
                                                    double (&up)[R], double (&se)[R])
{
    const double dt6 = dt / 6.0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double ur = (j == R - 1) ? uR : u[j + 1];
        const double w = u[j] + ur;
        const double dif = ur - u[j];
        const double aoff = gm.h6[j + 1] - gm.eh[j + 1];
        up[j] = __builtin_fma(dt6, w + u[j], aoff);
        if (j + 1 < R) lo[j + 1] = __builtin_fma(-dt6, w + ur, aoff);
        const double mx = fmax(fabs(w), 2.0e-10);
        const double t = __builtin_fma(w, dif, -hfs[j]);
        se[j] = t * BG_RCP(mx);
    }
    lo[0] = __builtin_fma(-dt6, __builtin_fma(2.0, u[0], uL), gm.h6[0] - gm.eh[0]);
}

template <int R>
__device__ __forceinline__ void assemble_general_p2(const ElemGeom<R>& gm, double dt, double kap, int N, int row0,
                                                    bool first, double mu1, const double (&u)[R], double uL,
                                                    double uR, double seL, const double (&g)[R],
                                                    const double (&se)[R], double (&lo)[R], double (&di)[R],
                                                    double (&up)[R], double (&rhs)[R])
{
    const double dt6 = dt / 6.0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double um = (j == 0) ? uL : u[j - 1];
        const double ur = (j == R - 1) ? uR : u[j + 1];
        const double sm = (j == 0) ? seL : se[j - 1];
        const int i = row0 + j;
        const double ddl = __builtin_fma(2.0, gm.h6[j], gm.eh[j]);          // h/3 + dt E/h of the left element
        const double ddr = __builtin_fma(2.0, gm.h6[j + 1], gm.eh[j + 1]);  // ... of the right element
        double d = __builtin_fma(dt6, um - ur, ddl + ddr);
        double b = __builtin_fma(-kap, sm, g[j]);
        double bb = __builtin_fma(kap, se[j], b);
        double l = lo[j], p = up[j];
        const bool is_last = (i == N - 1);
        const double dl = __builtin_fma(dt6, __builtin_fma(2.0, u[j], um), ddl);
        d = is_last ? dl : d;
        p = is_last ? 0.0 : p;
        bb = is_last ? b : bb;
        const bool pad = i >= N;
        d = pad ? 1.0 : d;
        p = pad ? 0.0 : p;
        l = pad ? 0.0 : l;
        bb = pad ? 0.0 : bb;
        if (j == 0) {
            d = first ? 1.0 : d;
            p = first ? 0.0 : p;
            l = first ? 0.0 : l;
            bb = first ? mu1 : bb;
        }
        double r = __builtin_fma(-l, um, bb);
        r = __builtin_fma(-d, u[j], r);
        r = __builtin_fma(-p, ur, r);
        lo[j] = l; di[j] = d; up[j] = p; rhs[j] = r;
    }
}

template <int R>
__device__ __forceinline__ void assemble_general(const ElemGeom<R>& gm, double dt, double kap, int N, int row0,
                                                 double mu1, const double (&u)[R], const double (&g)[R],
                                                 const double (&hfs)[R], double (&lo)[R], double (&di)[R],
                                                 double (&up)[R], double (&rhs)[R])
{
    const double uL = from_lane_below(u[R - 1]);
    const double uR = from_lane_above(u[0]);
    double se[R];
    assemble_general_p1<R>(gm, dt, u, uL, uR, hfs, lo, up, se);
    const double seL = from_lane_below(se[R - 1]);
    assemble_general_p2<R>(gm, dt, kap, N, row0, lane_id() == 0, mu1, u, uL, uR, seL, g, se, lo, di, up, rhs);
}

// One PCR step on normalised equations A x[j-s] + x[j] + C x[j+s] = D; neighbours come from
// DPP moves (CTRL_DN: from lower lanes, CTRL_UP: from higher lanes, applied REPS times, zero
// filled out of range, which is the identity equation).  LAST skips the A/C update.
template <int CTRL_DN, int CTRL_UP, int REPS, bool LAST>
__device__ __forceinline__ void pcr_step(double& A, double& C, double& D)
{
    const double Dm = dpp_shift<CTRL_DN, REPS>(D), Dp = dpp_shift<CTRL_UP, REPS>(D);
    const double Cm = dpp_shift<CTRL_DN, REPS>(C), Ap = dpp_shift<CTRL_UP, REPS>(A);
    double Bn = __builtin_fma(-Cm, A, 1.0);
    Bn = __builtin_fma(-Ap, C, Bn);
    double Dn = __builtin_fma(-Dm, A, D);
    Dn = __builtin_fma(-Dp, C, Dn);
    const double rb = BG_RCP(Bn);
    D = Dn * rb;
    if (!LAST) {
        const double Am = dpp_shift<CTRL_DN, REPS>(A), Cp = dpp_shift<CTRL_UP, REPS>(C);
        A = -(Am * A) * rb;
        C = -(Cp * C) * rb;
    }
}

// ---- Wang partition inside one lane, in pieces (shared by the one-wave and the workgroup-wide solver) ----
// wang_reduce: phase 1 eliminates the sub-diagonal downwards (lo[] becomes the left spike f[], di[] becomes
// 1/pivot), phase 2 the super-diagonal upwards from row R-3 (gs[] is the right spike).  Returns the last pivot.
template <int R>
__device__ __forceinline__ double wang_reduce(double (&lo)[R], double (&di)[R], const double (&up)[R],
                                              double (&rhs)[R], double (&gs)[R])
{
    static_assert(R > 1, "R == 1 has no interior rows");
    double dp = di[0];
    di[0] = BG_RCP(dp);
#pragma unroll
    for (int j = 1; j < R; ++j) {
        const double m = lo[j] * di[j - 1];
        dp = __builtin_fma(-m, up[j - 1], di[j]);
        di[j] = BG_RCP(dp);
        rhs[j] = __builtin_fma(-m, rhs[j - 1], rhs[j]);
        lo[j] = -m * lo[j - 1];
    }
    gs[R - 2] = up[R - 2];
#pragma unroll
    for (int j = R - 3; j >= 0; --j) {
        const double t = up[j] * di[j + 1];
        rhs[j] = __builtin_fma(-t, rhs[j + 1], rhs[j]);
        lo[j] = __builtin_fma(-t, lo[j + 1], lo[j]);
        gs[j] = -t * gs[j + 1];
    }
    return dp;
}

// Interface equation A x[p-1] + x[p] + C x[p+1] = D of this lane's last row, closed with the next lane's
// normalised row 0 (F0, G0, R0 = lo[0]/di[0], gs[0]/di[0], rhs[0]/di[0] over there; zeros past the last lane).
template <int R>
__device__ __forceinline__ void wang_interface(const double (&lo)[R], const double (&up)[R], const double (&rhs)[R],
                                               double dp, double F0, double G0, double R0, double& A, double& C,
                                               double& D)
{
    const double ul = up[R - 1];
    const double B = __builtin_fma(-ul, F0, dp);
    const double rb = BG_RCP(B);
    A = lo[R - 1] * rb;
    C = -(ul * G0) * rb;
    D = __builtin_fma(-ul, R0, rhs[R - 1]) * rb;
}

// Back-substitution: X = this lane's last unknown, XL = the previous lane's.  Solution lands in rhs[].
template <int R>
__device__ __forceinline__ void wang_finish(const double (&lo)[R], const double (&di)[R], double (&rhs)[R],
                                            const double (&gs)[R], double X, double XL)
{
#pragma unroll
    for (int j = 0; j < R - 1; ++j) {
        double v = __builtin_fma(-lo[j], XL, rhs[j]);
        v = __builtin_fma(-gs[j], X, v);
        rhs[j] = v * di[j];
    }
    rhs[R - 1] = X;
}

// PCR over 64 normalised equations, one per lane; returns x of this lane's equation.
// ds_bpermute costs ~24 cycles per dword on gfx950 against ~6 for a DPP move, so: strides 1 and 2 use
// wave_shr/wave_shl:1 (once / twice, zero filled), then ONE lane transpose (lane' = 16*(j&3) + (j>>2)) turns
// the four interleaved stride-4 systems into the four 16-lane DPP rows, where strides 4,8,16,32 are
// row_shr/shl 1,2,4,8 with zero fill at the row ends (= the system ends).
__device__ __forceinline__ double pcr64(double A, double C, double D)
{
    const int lane = lane_id();
    pcr_step<0x138, 0x130, 1, false>(A, C, D);
    pcr_step<0x138, 0x130, 2, false>(A, C, D);
    {
        const int src = ((4 * (lane & 15) + (lane >> 4)) << 2);
        A = from_lane_rot(A, src); C = from_lane_rot(C, src); D = from_lane_rot(D, src);
    }
    pcr_step<0x111, 0x101, 1, false>(A, C, D);
    pcr_step<0x112, 0x102, 1, false>(A, C, D);
    pcr_step<0x114, 0x104, 1, false>(A, C, D);
    pcr_step<0x118, 0x108, 1, true>(A, C, D);
    return from_lane_rot(D, (16 * (lane & 3) + (lane >> 2)) << 2);
}

// Pivot-free tridiagonal solve across the wave (Wang partition + PCR on the 64
// interface unknowns).  In: lo/di/up/rhs.  Out: solution in rhs.  lo, di are clobbered.
template <int R>
__device__ __forceinline__ void tridiag_solve(double (&lo)[R], double (&di)[R], const double (&up)[R],
                                              double (&rhs)[R])
{
    if constexpr (R == 1) {
        const double rb = BG_RCP(di[0]);
        rhs[0] = pcr64(lo[0] * rb, up[0] * rb, rhs[0] * rb);
    } else {
        double gs[R];
        const double dp = wang_reduce<R>(lo, di, up, rhs, gs);
        const double F0 = from_lane_above(lo[0] * di[0]);
        const double G0 = from_lane_above(gs[0] * di[0]);
        const double R0 = from_lane_above(rhs[0] * di[0]);
        double A, C, D;           // normalised interface equation: A x[p-1] + x[p] + C x[p+1] = D
        wang_interface<R>(lo, up, rhs, dp, F0, G0, R0, A, C, D);
        const double X = pcr64(A, C, D);
        wang_finish<R>(lo, di, rhs, gs, X, from_lane_below(X));
    }
}

}  // namespace bg
