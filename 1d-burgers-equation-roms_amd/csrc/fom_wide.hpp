// fom_wide.hpp -- one WORKGROUP (4 wavefronts) per sample, for meshes beyond one wavefront's
// registers (1536 < N <= 8192).  Same arithmetic as fom_device.hpp; what changes is where the
// halo values come from and how the 256 interface equations of the Wang partition are solved.
//
// Row distribution: thread g = 64*wave + lane of the workgroup owns the R consecutive rows
// [g*R, g*R + R), N <= 256*R.  Neighbour values travel through LDS arrays indexed by g with two
// zero pads on either side (the zero fill the DPP shifts give inside one wave).
//
// Interface system (256 normalised equations, one per thread): parallel cyclic reduction.  The
// equations are re-dealt so that equation e sits in wave e & 3, lane e >> 2; the strides 1 and 2
// then cross waves (two LDS exchanges), after which the four interleaved stride-4 systems each
// live in one wave in lane order and finish with the in-wave pcr64 of fom_device.hpp.
#pragma once
#include "fom_device.hpp"

namespace bg {

constexpr int WIDE_WAVES = 4;
constexpr int WIDE_THREADS = 64 * WIDE_WAVES;
constexpr int WIDE_PAD = 2;
constexpr int WIDE_LEN = WIDE_THREADS + 2 * WIDE_PAD;

struct WideLds {
    double ufirst[WIDE_LEN], ulast[WIDE_LEN];   // u of every thread's first / last row
    double se[WIDE_LEN];                        // SUPG term of every thread's last element
    double fgr[3][WIDE_LEN];                    // normalised row 0 of every thread (F0, G0, R0)
    double pcr[2][3][WIDE_LEN];                 // interface equations (A, C, D), double-buffered
    double x[WIDE_LEN];                         // interface unknowns
    double nrm[2][WIDE_WAVES];                  // per-wave partial norms
};

// pads are written once; every later store goes to [WIDE_PAD, WIDE_PAD + 256)
__device__ __forceinline__ void wide_init(WideLds& s, int tid)
{
    double* p = reinterpret_cast<double*>(&s);
    for (int i = tid; i < (int)(sizeof(WideLds) / sizeof(double)); i += WIDE_THREADS) p[i] = 0.0;
    __syncthreads();
}

// u of the row below this thread's first row and above its last row (0 outside the mesh)
template <int R>
__device__ __forceinline__ void wide_halo_u(WideLds& s, int g, const double (&u)[R], double& uL, double& uR)
{
    s.ufirst[g + WIDE_PAD] = u[0];
    s.ulast[g + WIDE_PAD] = u[R - 1];
    __syncthreads();
    uL = s.ulast[g + WIDE_PAD - 1];
    uR = s.ufirst[g + WIDE_PAD + 1];
}

template <int R>
__device__ __forceinline__ void wide_mass_rhs(WideLds& s, int g, const ElemGeom<R>& gm, int N, const double (&u)[R],
                                              const double (&fdt)[R], double (&gv)[R])
{
    double uL, uR;
    wide_halo_u<R>(s, g, u, uL, uR);
    mass_rhs_general_core<R>(gm, N, g * R, u, uL, uR, fdt, gv);
    __syncthreads();              // the assembly that follows refills ufirst/ulast
}

template <int R>
__device__ __forceinline__ void wide_assemble(WideLds& s, int g, const ElemGeom<R>& gm, double dt, double kap, int N,
                                              double mu1, const double (&u)[R], const double (&gv)[R],
                                              const double (&hfs)[R], double (&lo)[R], double (&di)[R],
                                              double (&up)[R], double (&rhs)[R])
{
    double uL, uR, se[R];
    wide_halo_u<R>(s, g, u, uL, uR);
    assemble_general_p1<R>(gm, dt, u, uL, uR, hfs, lo, up, se);
    s.se[g + WIDE_PAD] = se[R - 1];
    __syncthreads();
    const double seL = s.se[g + WIDE_PAD - 1];
    assemble_general_p2<R>(gm, dt, kap, N, g * R, g == 0, mu1, u, uL, uR, seL, gv, se, lo, di, up, rhs);
}

// Parked rows: at 32 rows per thread the per-row constants of a whole time step, g = M u^n + dt F and hfs, no longer fit
// the register file beside the solver's working set (404 bytes of scratch per thread).  The uniform-mesh N > 6144 kernel
// keeps them in LDS instead, row-major over threads (park[j][g]: consecutive threads, consecutive addresses) and reads
// each value where the assembly uses it.
template <int R>
struct WidePark {
    double g[R][WIDE_THREADS];
    double hfs[R][WIDE_THREADS];
};

template <int R>
__device__ __forceinline__ void wide_mass_rhs_uni_parked(WideLds& s, WidePark<R>& pk, int g, const MeshConst& c, int N,
                                                         const double (&u)[R], const double (&fdt)[R])
{
    double uL, uR;
    wide_halo_u<R>(s, g, u, uL, uR);
    mass_rhs_core_to<R, false>(c, N, g * R, u, uL, uR, fdt, [&](int j, double v) { pk.g[j][g] = v; });
    __syncthreads();
}

template <int R>
__device__ __forceinline__ void wide_assemble_uni_parked(WideLds& s, const WidePark<R>& pk, int g, const MeshConst& c, int N,
                                                         double mu1, const double (&u)[R], double (&lo)[R], double (&di)[R],
                                                         double (&up)[R], double (&rhs)[R])
{
    double uL, uR, se[R];
    wide_halo_u<R>(s, g, u, uL, uR);
    assemble_p1_from<R>(c, u, uL, uR, [&](int j) { return pk.hfs[j][g]; }, lo, up, se);
    s.se[g + WIDE_PAD] = se[R - 1];
    __syncthreads();
    const double seL = s.se[g + WIDE_PAD - 1];
    assemble_p2_from<R, false>(c, N, g * R, g == 0, false, mu1, u, uL, uR, seL, [&](int j) { return pk.g[j][g]; }, se, lo,
                               di, up, rhs);
}

// uniform-mesh variants (one element length: no per-element registers, fewer instructions)
template <int R>
__device__ __forceinline__ void wide_mass_rhs_uni(WideLds& s, int g, const MeshConst& c, int N, const double (&u)[R],
                                                  const double (&fdt)[R], double (&gv)[R])
{
    double uL, uR;
    wide_halo_u<R>(s, g, u, uL, uR);
    mass_rhs_core<R, false>(c, N, g * R, u, uL, uR, fdt, gv);
    __syncthreads();
}

template <int R>
__device__ __forceinline__ void wide_assemble_uni(WideLds& s, int g, const MeshConst& c, int N, double mu1,
                                                  const double (&u)[R], const double (&gv)[R],
                                                  const double (&hfs)[R], double (&lo)[R], double (&di)[R],
                                                  double (&up)[R], double (&rhs)[R])
{
    double uL, uR, se[R];
    wide_halo_u<R>(s, g, u, uL, uR);
    assemble_p1<R>(c, u, uL, uR, hfs, lo, up, se);
    s.se[g + WIDE_PAD] = se[R - 1];
    __syncthreads();
    const double seL = s.se[g + WIDE_PAD - 1];
    assemble_p2<R, false>(c, N, g * R, g == 0, false, mu1, u, uL, uR, seL, gv, se, lo, di, up, rhs);
}

// One PCR step of stride S on equation e, neighbours from the LDS copy `buf` of all equations.
template <int S>
__device__ __forceinline__ void wide_pcr_step(const double (&buf)[3][WIDE_LEN], int e, double& A, double& C, double& D)
{
    const int m = e + WIDE_PAD - S, p = e + WIDE_PAD + S;
    const double Am = buf[0][m], Cm = buf[1][m], Dm = buf[2][m];
    const double Ap = buf[0][p], Cp = buf[1][p], Dp = buf[2][p];
    double Bn = __builtin_fma(-Cm, A, 1.0);
    Bn = __builtin_fma(-Ap, C, Bn);
    double Dn = __builtin_fma(-Dm, A, D);
    Dn = __builtin_fma(-Dp, C, Dn);
    const double rb = BG_RCP(Bn);
    D = Dn * rb;
    A = -(Am * A) * rb;
    C = -(Cp * C) * rb;
}

// Pivot-free tridiagonal solve across the workgroup.  In: lo/di/up/rhs.  Out: solution in rhs.
template <int R>
__device__ __forceinline__ void wide_tridiag_solve(WideLds& s, int g, double (&lo)[R], double (&di)[R],
                                                   const double (&up)[R], double (&rhs)[R])
{
    double gs[R];
    const double dp = wang_reduce<R>(lo, di, up, rhs, gs);
    s.fgr[0][g + WIDE_PAD] = lo[0] * di[0];
    s.fgr[1][g + WIDE_PAD] = gs[0] * di[0];
    s.fgr[2][g + WIDE_PAD] = rhs[0] * di[0];
    __syncthreads();
    double A, C, D;
    wang_interface<R>(lo, up, rhs, dp, s.fgr[0][g + WIDE_PAD + 1], s.fgr[1][g + WIDE_PAD + 1],
                      s.fgr[2][g + WIDE_PAD + 1], A, C, D);
    s.pcr[0][0][g + WIDE_PAD] = A;
    s.pcr[0][1][g + WIDE_PAD] = C;
    s.pcr[0][2][g + WIDE_PAD] = D;
    __syncthreads();
    const int e = (g & 63) * WIDE_WAVES + (g >> 6);          // the equation this thread carries through the PCR
    A = s.pcr[0][0][e + WIDE_PAD]; C = s.pcr[0][1][e + WIDE_PAD]; D = s.pcr[0][2][e + WIDE_PAD];
    wide_pcr_step<1>(s.pcr[0], e, A, C, D);
    s.pcr[1][0][e + WIDE_PAD] = A;
    s.pcr[1][1][e + WIDE_PAD] = C;
    s.pcr[1][2][e + WIDE_PAD] = D;
    __syncthreads();
    wide_pcr_step<2>(s.pcr[1], e, A, C, D);
    s.x[e + WIDE_PAD] = pcr64(A, C, D);                      // stride-4 system e & 3 = this wave, in lane order
    __syncthreads();
    wang_finish<R>(lo, di, rhs, gs, s.x[g + WIDE_PAD], s.x[g + WIDE_PAD - 1]);
}

// sums over the whole workgroup, the same value in every thread (fixed summation order)
__device__ __forceinline__ void wide_sum2(WideLds& s, int g, double a, double b, double& sa, double& sb)
{
    wave_sum2(a, b, a, b);
    if ((g & 63) == 0) { s.nrm[0][g >> 6] = a; s.nrm[1][g >> 6] = b; }
    __syncthreads();
    sa = (s.nrm[0][0] + s.nrm[0][1]) + (s.nrm[0][2] + s.nrm[0][3]);
    sb = (s.nrm[1][0] + s.nrm[1][1]) + (s.nrm[1][2] + s.nrm[1][3]);
}

}  // namespace bg
