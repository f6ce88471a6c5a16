// fd.hip -- fused batched finite-difference true-Newton stepper (widening row f.4 of SURVEY 8).
//
// Replaces FDBurgers.fom_burgers_newton with the analytical Jacobian (reference
// FD/fd_burgers.py:59-107; residual :28-35, Jacobian :37-44, boundary values :19-22).
// Same skeleton as the FEM kernel: one wavefront per (mu1, mu2) sample for the whole time loop,
// rows in registers, the tridiagonal Newton system solved by the Wang + PCR solver of
// fom_device.hpp.  The Jacobian is diagonally dominant here (1/dt + 2 nu/dx^2 on the diagonal),
// so pivot-free elimination is safe.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "fom_device.hpp"
#include "fom_wide.hpp"

namespace {

using namespace bg;

struct FdArgs {
    const double* x;
    const double* u0;
    const double* mu1;
    const double* mu2;
    double* hist;
    int32_t* iters;
    int32_t* flags;
    double dt, tol;
    int N, B, nsteps, max_it;
};

__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_mov<0x111>(v));
    v = fmax(v, dpp_mov<0x112>(v));
    v = fmax(v, dpp_mov<0x114>(v));
    v = fmax(v, dpp_mov<0x118>(v));
    v = fmax(v, dpp_mov<0x142, 0xA>(v));
    v = fmax(v, dpp_mov<0x143, 0xC>(v));
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Where neighbour values, maxima and the tridiagonal solve come from: one wavefront per sample (DPP inside the
// wave) or one 256-thread workgroup per sample (LDS across its four waves, fom_wide.hpp).
struct WaveTopo {
    int g;                                           // index of this thread among the sample's threads (= lane)
    __device__ __forceinline__ double below(double v) { return from_lane_below(v); }
    __device__ __forceinline__ double above(double v) { return from_lane_above(v); }
    __device__ __forceinline__ double gmax(double v) { return wave_max(v); }
    template <int R>
    __device__ __forceinline__ void solve(double (&lo)[R], double (&di)[R], const double (&up)[R], double (&rhs)[R])
    {
        tridiag_solve<R>(lo, di, up, rhs);
    }
};

struct WgTopo {
    WideLds& s;
    int g;
    int par;                                         // alternating exchange buffers: one barrier per exchange
    __device__ __forceinline__ double shift(double v, int d)
    {
        double* buf = par ? s.ulast : s.ufirst;
        par ^= 1;
        buf[g + WIDE_PAD] = v;
        __syncthreads();
        return buf[g + WIDE_PAD + d];
    }
    __device__ __forceinline__ double below(double v) { return shift(v, -1); }
    __device__ __forceinline__ double above(double v) { return shift(v, +1); }
    __device__ __forceinline__ double gmax(double v)
    {
        v = wave_max(v);
        double* buf = s.nrm[par];
        par ^= 1;
        if ((g & 63) == 0) buf[g >> 6] = v;
        __syncthreads();
        return fmax(fmax(buf[0], buf[1]), fmax(buf[2], buf[3]));
    }
    template <int R>
    __device__ __forceinline__ void solve(double (&lo)[R], double (&di)[R], const double (&up)[R], double (&rhs)[R])
    {
        wide_tridiag_solve<R>(s, g, lo, di, up, rhs);
    }
};

// boundary values of the reference's apply_dirichlet_bc: U[0] = mu1, U[-1] = U[-2]
template <int R, class Topo>
__device__ __forceinline__ void apply_bc(Topo& tp, double (&u)[R], int N, int row0, double mu1)
{
    const double uL = tp.below(u[R - 1]);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int i = row0 + j;
        const double um = (j == 0) ? uL : u[j - 1];
        u[j] = (i == 0) ? mu1 : ((i == N - 1) ? um : u[j]);
    }
}

// the whole time loop of sample s; every branch below is uniform over the sample's threads
template <int R, class Topo>
__device__ __forceinline__ void fd_sample(const FdArgs& a, Topo& tp, int s)
{
    const int N = a.N, row0 = tp.g * R;
    const double dx = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const double idt = 1.0 / a.dt, i2dx = 1.0 / (2.0 * dx), idx2 = 1.0 / (dx * dx);
    const double mu1 = a.mu1[s], mu2 = a.mu2[s];
    double u[R], src[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int i = row0 + j;
        u[j] = (i < N) ? a.u0[(size_t)s * N + i] : 0.0;
        src[j] = (i < N) ? 0.02 * exp(mu2 * a.x[i]) : 0.0;
    }
    apply_bc<R>(tp, u, N, row0, mu1);
    double* hist = a.hist + (size_t)s * (size_t)(a.nsteps + 1) * (size_t)N;
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (row0 + j < N) hist[row0 + j] = u[j];

    int flags = 0;
    for (int step = 0; step < a.nsteps; ++step) {
        double uprev[R];
#pragma unroll
        for (int j = 0; j < R; ++j) uprev[j] = u[j];
        int k = 0;
        bool converged = false;
        for (int it = 0; it < a.max_it; ++it) {
            apply_bc<R>(tp, u, N, row0, mu1);
            const double uL = tp.below(u[R - 1]);
            const double uR = tp.above(u[0]);
            double mloc = 0.0;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int i = row0 + j;
                mloc = (i >= 1 && i <= N - 2) ? fmax(mloc, fabs(u[j])) : mloc;
            }
            const double umax_int = tp.gmax(mloc);                  // max |U_guess[1:-1]|
            const double nu = 0.25 * dx * fmax(umax_int, fabs(mu1));  // max over ALL entries (U[-1] = U[-2])
            const double nud = nu * idx2;
            double lo[R], di[R], up[R], rhs[R];
            double rloc = 0.0;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int i = row0 + j;
                const double um = (j == 0) ? uL : u[j - 1];
                const double ur = (j == R - 1) ? uR : u[j + 1];
                const bool interior = (i >= 1) && (i <= N - 2);
                const double conv = (0.5 * ur * ur - 0.5 * um * um) * i2dx;
                const double diff = nu * ((ur - 2.0 * u[j]) + um) * idx2;
                const double Ri = (u[j] - uprev[j]) * idt + conv - src[j] - diff;
                rloc = interior ? fmax(rloc, fabs(Ri)) : rloc;
                lo[j] = (interior && i > 1) ? (-um * i2dx - nud) : 0.0;
                up[j] = (interior && i < N - 2) ? (ur * i2dx - nud) : 0.0;
                di[j] = interior ? (idt + 2.0 * nud) : 1.0;
                rhs[j] = interior ? -Ri : 0.0;
            }
            const double res = tp.gmax(rloc);
            if (res < a.tol) { converged = true; break; }
            tp.template solve<R>(lo, di, up, rhs);
            double dloc = 0.0;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                dloc = fmax(dloc, fabs(rhs[j]));
                u[j] += rhs[j];
            }
            const double rel = tp.gmax(dloc) / fmax(umax_int, 1e-15);
            ++k;
            if (!(rel - rel == 0.0)) flags |= BG_FLAG_NONFINITE;
            if (rel < a.tol) { converged = true; break; }
        }
        if (!converged) flags |= BG_FLAG_HIT_CAP;
        apply_bc<R>(tp, u, N, row0, mu1);
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (row0 + j < N) hist[(size_t)(step + 1) * N + row0 + j] = u[j];
        if (tp.g == 0) a.iters[(size_t)s * a.nsteps + step] = k;
    }
    if (tp.g == 0) a.flags[s] = flags;
}

template <int R>
__global__ __launch_bounds__(256, 1) void fd_fused_kernel(FdArgs a)
{
    const int s = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (s >= a.B) return;                                           // wave-uniform
    WaveTopo tp{lane_id()};
    fd_sample<R>(a, tp, s);
}

template <int R>
__global__ __launch_bounds__(WIDE_THREADS, 1) void fd_wide_kernel(FdArgs a)
{
    __shared__ WideLds lds;
    wide_init(lds, threadIdx.x);
    WgTopo tp{lds, (int)threadIdx.x, 0};
    fd_sample<R>(a, tp, blockIdx.x);
}

}  // namespace

extern "C" int bg_fd_run(int N, int B, int nsteps, const double* x, const double* u0, const double* mu1,
                         const double* mu2, double dt, double tol, int max_it, double* hist, int32_t* iters,
                         int32_t* flags, void* stream)
{
    if (N < 3 || B < 0 || nsteps < 0 || max_it < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!x || !u0 || !mu1 || !mu2 || !hist || !flags || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    if (N > WIDE_THREADS * 32) return BG_ERR_UNSUPPORTED_N;
    FdArgs a{x, u0, mu1, mu2, hist, iters, flags, dt, tol, N, B, nsteps, max_it};
    const dim3 grid((B + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (N > 2048) {                                  // one workgroup per sample
        const int rw = (N + WIDE_THREADS - 1) / WIDE_THREADS;
        if (rw <= 12) hipLaunchKernelGGL((fd_wide_kernel<12>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
        else if (rw <= 16) hipLaunchKernelGGL((fd_wide_kernel<16>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
        else if (rw <= 24) hipLaunchKernelGGL((fd_wide_kernel<24>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
        else hipLaunchKernelGGL((fd_wide_kernel<32>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
        return bg::check_launch();
    }
    const int r = (N + 63) / 64;
#define BG_FD(RV) hipLaunchKernelGGL((fd_fused_kernel<RV>), grid, block, 0, st, a)
    if (r <= 1) BG_FD(1);
    else if (r <= 2) BG_FD(2);
    else if (r <= 4) BG_FD(4);
    else if (r <= 8) BG_FD(8);
    else if (r <= 12) BG_FD(12);
    else if (r <= 16) BG_FD(16);
    else if (r <= 24) BG_FD(24);
    else BG_FD(32);
#undef BG_FD
    return bg::check_launch();
}
