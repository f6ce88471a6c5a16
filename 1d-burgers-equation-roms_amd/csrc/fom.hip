// fom.hip -- fused batched FOM time-stepper for gfx950 and its C-ABI entry points.
//
// Replaces the body of FEMBurgers.fom_burgers (reference FEM/fem_burgers.py:646-707):
// one wavefront integrates one (mu1, mu2) sample through all time steps and all Picard
// iterations; the state never leaves registers between iterations, HBM sees one
// coalesced row write per time step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "fom_device.hpp"
#include "fom_wide.hpp"

namespace bg {
thread_local int tls_last_hip_error = 0;
}

namespace {

using namespace bg;

struct FomArgs {
    const double* x;
    const double* u0;
    const double* mu1;
    const double* mu2;
    double* hist;
    int32_t* iters;
    int32_t* flags;
    double dt, E, tol2;
    int N, B, nsteps, max_it, supg;
    double* errs;        // [B][nsteps][max_it], TRACE instantiations only: the error of every Picard iteration
};

constexpr int WAVES_PER_WG = 4;

template <int R>
__device__ __forceinline__ void load_rows(const double* __restrict__ src, int N, int row0, bool full,
                                          double (&u)[R])
{
#pragma unroll
    for (int j = 0; j < R; ++j) u[j] = (full || row0 + j < N) ? src[row0 + j] : 0.0;
}

template <int R>
__device__ __forceinline__ void store_rows(double* __restrict__ dst, int N, int row0, bool full,
                                           const double (&u)[R])
{
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (full || row0 + j < N) dst[row0 + j] = u[j];
}

// TRACE: also store error_U = ||dU|| / ||U1|| of every iteration (what the reference prints at :664); a separate
// instantiation, so that the hot kernel carries no trace of it.
template <int R, bool FULL, bool UNI = true, bool TRACE = false>
__global__ __launch_bounds__(64 * WAVES_PER_WG, 1) void fom_fused_kernel(FomArgs a)
{
    const int lane = lane_id();
    const int s = blockIdx.x * WAVES_PER_WG + (int)(threadIdx.x >> 6);
    if (s >= a.B) return;                                   // wave-uniform
    const int N = a.N;
    const int row0 = lane * R;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const MeshConst c = make_mesh_const(h, a.dt, a.E, a.supg);
    const double mu1 = a.mu1[s], mu2 = a.mu2[s];

    double hfs[R], fdt[R], u[R], g[R];
    ElemGeom<UNI ? 0 : R> gm;
    if constexpr (UNI) {
        forcing_setup<R>(a.x, N, row0, mu2, c.h, a.dt, hfs, fdt);
    } else {
        geom_setup<R>(a.x, N, row0, a.dt, a.E, gm);
        forcing_setup_general<R>(a.x, N, row0, mu2, a.dt, hfs, fdt);
    }

    double* hist = a.hist + (size_t)s * (size_t)(a.nsteps + 1) * (size_t)N;
    load_rows<R>(a.u0 + (size_t)s * N, N, row0, FULL, u);
    store_rows<R>(hist, N, row0, FULL, u);

    int flags = 0;
    for (int step = 0; step < a.nsteps; ++step) {
        if constexpr (UNI) mass_rhs<R, FULL>(c, N, row0, u, fdt, g);
        else mass_rhs_general<R>(gm, N, row0, u, fdt, g);
        int k = 0;
        bool more;
        do {
            double lo[R], di[R], up[R], rhs[R];
            if constexpr (UNI) assemble<R, FULL>(c, N, row0, mu1, u, g, hfs, lo, di, up, rhs);
            else assemble_general<R>(gm, a.dt, c.kap, N, row0, mu1, u, g, hfs, lo, di, up, rhs);
            tridiag_solve<R>(lo, di, up, rhs);
            double nd = 0.0, nu = 0.0;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                u[j] += rhs[j];
                nd = __builtin_fma(rhs[j], rhs[j], nd);
                nu = __builtin_fma(u[j], u[j], nu);
            }
            wave_sum2(nd, nu, nd, nu);
            if constexpr (TRACE) {
                if (lane == 0) a.errs[((size_t)s * a.nsteps + step) * a.max_it + k] = sqrt(nd) / sqrt(nu);
            }
            ++k;
            // reference: error = ||dU|| / ||U1||; continue while error > tol and k < cap.
            // NaN compares false and ends the loop, as in the reference.
            more = (nd > a.tol2 * nu) && (k < a.max_it);
            if (!(nd - nd == 0.0) || !(nu - nu == 0.0)) flags |= BG_FLAG_NONFINITE;
        } while (more);
        if (k >= a.max_it) flags |= BG_FLAG_HIT_CAP;
        store_rows<R>(hist + (size_t)(step + 1) * N, N, row0, FULL, u);
        if (lane == 0) a.iters[(size_t)s * a.nsteps + step] = k;
    }
    if (lane == 0) a.flags[s] = flags;
}

// ---- diagnostics kernels: one assembly / one solve ---------------------------------
struct AsmArgs {
    const double *x, *uk, *un, *mu1, *mu2;
    double *lo, *di, *up, *rhs;
    double dt, E;
    int N, B, supg;
};

template <int R, bool FULL, bool UNI = true>
__global__ __launch_bounds__(64 * WAVES_PER_WG, 1) void fom_assemble_kernel(AsmArgs a)
{
    const int lane = lane_id();
    const int s = blockIdx.x * WAVES_PER_WG + (int)(threadIdx.x >> 6);
    if (s >= a.B) return;
    const int N = a.N, row0 = lane * R;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const MeshConst c = make_mesh_const(h, a.dt, a.E, a.supg);
    double hfs[R], fdt[R], u[R], un[R], g[R], lo[R], di[R], up[R], rhs[R];
    load_rows<R>(a.un + (size_t)s * N, N, row0, FULL, un);
    load_rows<R>(a.uk + (size_t)s * N, N, row0, FULL, u);
    if constexpr (UNI) {
        forcing_setup<R>(a.x, N, row0, a.mu2[s], c.h, a.dt, hfs, fdt);
        mass_rhs<R, FULL>(c, N, row0, un, fdt, g);
        assemble<R, FULL>(c, N, row0, a.mu1[s], u, g, hfs, lo, di, up, rhs);
    } else {
        ElemGeom<R> gm;
        geom_setup<R>(a.x, N, row0, a.dt, a.E, gm);
        forcing_setup_general<R>(a.x, N, row0, a.mu2[s], a.dt, hfs, fdt);
        mass_rhs_general<R>(gm, N, row0, un, fdt, g);
        assemble_general<R>(gm, a.dt, c.kap, N, row0, a.mu1[s], u, g, hfs, lo, di, up, rhs);
    }
    store_rows<R>(a.lo + (size_t)s * N, N, row0, FULL, lo);
    store_rows<R>(a.di + (size_t)s * N, N, row0, FULL, di);
    store_rows<R>(a.up + (size_t)s * N, N, row0, FULL, up);
    store_rows<R>(a.rhs + (size_t)s * N, N, row0, FULL, rhs);
}

struct SolveArgs {
    const double *lo, *di, *up, *rhs;
    double* sol;
    int N, B;
};

template <int R>
__global__ __launch_bounds__(64 * WAVES_PER_WG, 1) void tridiag_solve_kernel(SolveArgs a)
{
    const int lane = lane_id();
    const int s = blockIdx.x * WAVES_PER_WG + (int)(threadIdx.x >> 6);
    if (s >= a.B) return;
    const int N = a.N, row0 = lane * R;
    double lo[R], di[R], up[R], rhs[R];
    const size_t off = (size_t)s * N;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool in = row0 + j < N;
        lo[j] = in ? a.lo[off + row0 + j] : 0.0;
        di[j] = in ? a.di[off + row0 + j] : 1.0;
        up[j] = in ? a.up[off + row0 + j] : 0.0;
        rhs[j] = in ? a.rhs[off + row0 + j] : 0.0;
    }
    tridiag_solve<R>(lo, di, up, rhs);
    store_rows<R>(a.sol + off, N, row0, false, rhs);
}

// ---- workgroup-per-sample variants (2048 < N <= 8192), see fom_wide.hpp ---------------------
// PARK (uniform mesh, 32 rows per thread): g and hfs live in LDS (WidePark) instead of 128 VGPRs per thread.
template <int R, bool UNI>
__global__ __launch_bounds__(WIDE_THREADS, 1) void fom_wide_kernel(FomArgs a)
{
    constexpr bool PARK = UNI && R >= 32;
    __shared__ WideLds lds;
    __shared__ WidePark<PARK ? R : 1> park;
    const int g = threadIdx.x, s = blockIdx.x;              // one workgroup per sample
    const int N = a.N, row0 = g * R;
    wide_init(lds, g);
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const MeshConst c = make_mesh_const(h, a.dt, a.E, a.supg);
    const double mu1 = a.mu1[s], mu2 = a.mu2[s];
    double hfs[PARK ? 1 : R], fdt[R], u[R], gv[PARK ? 1 : R];
    ElemGeom<UNI ? 0 : R> gm;
    if constexpr (PARK) {
        double hf[R];
        forcing_setup<R>(a.x, N, row0, mu2, c.h, a.dt, hf, fdt);
#pragma unroll
        for (int j = 0; j < R; ++j) park.hfs[j][g] = hf[j];
    } else if constexpr (UNI) {
        forcing_setup<R>(a.x, N, row0, mu2, c.h, a.dt, hfs, fdt);
    } else {
        geom_setup<R>(a.x, N, row0, a.dt, a.E, gm);
        forcing_setup_general<R>(a.x, N, row0, mu2, a.dt, hfs, fdt);
    }
    double* hist = a.hist + (size_t)s * (size_t)(a.nsteps + 1) * (size_t)N;
    load_rows<R>(a.u0 + (size_t)s * N, N, row0, false, u);
    store_rows<R>(hist, N, row0, false, u);
    int flags = 0;
    for (int step = 0; step < a.nsteps; ++step) {
        if constexpr (PARK) wide_mass_rhs_uni_parked<R>(lds, park, g, c, N, u, fdt);
        else if constexpr (UNI) wide_mass_rhs_uni<R>(lds, g, c, N, u, fdt, gv);
        else wide_mass_rhs<R>(lds, g, gm, N, u, fdt, gv);
        int k = 0;
        bool more;                                          // workgroup-uniform: every thread sees the same sums
        do {
            double lo[R], di[R], up[R], rhs[R];
            if constexpr (PARK) wide_assemble_uni_parked<R>(lds, park, g, c, N, mu1, u, lo, di, up, rhs);
            else if constexpr (UNI) wide_assemble_uni<R>(lds, g, c, N, mu1, u, gv, hfs, lo, di, up, rhs);
            else wide_assemble<R>(lds, g, gm, a.dt, c.kap, N, mu1, u, gv, hfs, lo, di, up, rhs);
            wide_tridiag_solve<R>(lds, g, lo, di, up, rhs);
            double nd = 0.0, nu = 0.0;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                u[j] += rhs[j];
                nd = __builtin_fma(rhs[j], rhs[j], nd);
                nu = __builtin_fma(u[j], u[j], nu);
            }
            wide_sum2(lds, g, nd, nu, nd, nu);
            ++k;
            more = (nd > a.tol2 * nu) && (k < a.max_it);
            if (!(nd - nd == 0.0) || !(nu - nu == 0.0)) flags |= BG_FLAG_NONFINITE;
        } while (more);
        if (k >= a.max_it) flags |= BG_FLAG_HIT_CAP;
        store_rows<R>(hist + (size_t)(step + 1) * N, N, row0, false, u);
        if (g == 0) a.iters[(size_t)s * a.nsteps + step] = k;
    }
    if (g == 0) a.flags[s] = flags;
}

template <int R, bool UNI>
__global__ __launch_bounds__(WIDE_THREADS, 1) void fom_assemble_wide_kernel(AsmArgs a)
{
    __shared__ WideLds lds;
    const int g = threadIdx.x, s = blockIdx.x;
    const int N = a.N, row0 = g * R;
    wide_init(lds, g);
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const MeshConst c = make_mesh_const(h, a.dt, a.E, a.supg);
    double hfs[R], fdt[R], u[R], un[R], gv[R], lo[R], di[R], up[R], rhs[R];
    load_rows<R>(a.un + (size_t)s * N, N, row0, false, un);
    load_rows<R>(a.uk + (size_t)s * N, N, row0, false, u);
    if constexpr (UNI) {
        forcing_setup<R>(a.x, N, row0, a.mu2[s], c.h, a.dt, hfs, fdt);
        wide_mass_rhs_uni<R>(lds, g, c, N, un, fdt, gv);
        wide_assemble_uni<R>(lds, g, c, N, a.mu1[s], u, gv, hfs, lo, di, up, rhs);
    } else {
        ElemGeom<R> gm;
        geom_setup<R>(a.x, N, row0, a.dt, a.E, gm);
        forcing_setup_general<R>(a.x, N, row0, a.mu2[s], a.dt, hfs, fdt);
        wide_mass_rhs<R>(lds, g, gm, N, un, fdt, gv);
        wide_assemble<R>(lds, g, gm, a.dt, c.kap, N, a.mu1[s], u, gv, hfs, lo, di, up, rhs);
    }
    store_rows<R>(a.lo + (size_t)s * N, N, row0, false, lo);
    store_rows<R>(a.di + (size_t)s * N, N, row0, false, di);
    store_rows<R>(a.up + (size_t)s * N, N, row0, false, up);
    store_rows<R>(a.rhs + (size_t)s * N, N, row0, false, rhs);
}

template <int R>
__global__ __launch_bounds__(WIDE_THREADS, 1) void tridiag_solve_wide_kernel(SolveArgs a)
{
    __shared__ WideLds lds;
    const int g = threadIdx.x, s = blockIdx.x;
    const int N = a.N, row0 = g * R;
    wide_init(lds, g);
    double lo[R], di[R], up[R], rhs[R];
    const size_t off = (size_t)s * N;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool in = row0 + j < N;
        lo[j] = in ? a.lo[off + row0 + j] : 0.0;
        di[j] = in ? a.di[off + row0 + j] : 1.0;
        up[j] = in ? a.up[off + row0 + j] : 0.0;
        rhs[j] = in ? a.rhs[off + row0 + j] : 0.0;
    }
    wide_tridiag_solve<R>(lds, g, lo, di, up, rhs);
    store_rows<R>(a.sol + off, N, row0, false, rhs);
}

// ---- batched transpose out[b][c][r] = in[b][r][c] ----------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in,
                                                        double* __restrict__ out, int nb, int rows, int cols)
{
    __shared__ double tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int b = blockIdx.z; b < nb; b += gridDim.z) {        // grid.z is capped at 65535
        const size_t base = (size_t)b * rows * cols;
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
            int r = r0 + ty + k, cc = c0 + tx;
            if (r < rows && cc < cols) tile[ty + k][tx] = in[base + (size_t)r * cols + cc];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
            int cc = c0 + ty + k, r = r0 + tx;
            if (r < rows && cc < cols) out[base + (size_t)cc * rows + r] = tile[tx][ty + k];
        }
        __syncthreads();
    }
}

// ---- dispatch ------------------------------------------------------------------------
constexpr int kRowsPerLane[] = {1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16, 24};

int rows_per_lane(int N)
{
    for (int r : kRowsPerLane)
        if (N <= 64 * r) return r;
    return 0;
}

// workgroup-per-sample kernels: 8, 12, 16, 24 or 32 rows per THREAD of a 256-thread workgroup
// Measured (tools/time_wide.py, B = 1024): at 32 rows per lane the wave-per-sample kernel spills (N = 2048:
// 1.1e8 steps/s, N = 1800: 3.4e7) and the workgroup kernel at 8 rows per thread wins (1.29e8 / 1.24e8); at 24
// rows per lane the wave kernel still wins (N = 1536: 2.7e8 against 1.4e8).
constexpr int kWaveMaxN = 64 * 24, kWideMaxN = WIDE_THREADS * 32;

// Instantiations at 2 or 4 rows per thread (round 3) put a sample of N <= 512, 1024 nodes on all four SIMDs of a CU instead
// of on one wavefront: built as the low-latency form for small batches (the strong-scaling share of a sweep, the
// reference's one-mu-per-call pattern, FEM/paper_training_stage.py:28-49), measured slower than the wavefront form
// (see bg_fom_run) and therefore opt-in (BG_OPT_FOM_WIDE).
template <typename F>
int dispatch_wide(int N, F&& f)
{
    if (N <= WIDE_THREADS * 2) return f(std::integral_constant<int, 2>{});     // (N <= 256: identity rows beyond N)
    if (N <= WIDE_THREADS * 4) return f(std::integral_constant<int, 4>{});
    if (N <= WIDE_THREADS * 8) return f(std::integral_constant<int, 8>{});
    if (N <= WIDE_THREADS * 12) return f(std::integral_constant<int, 12>{});
    if (N <= WIDE_THREADS * 16) return f(std::integral_constant<int, 16>{});
    if (N <= WIDE_THREADS * 24) return f(std::integral_constant<int, 24>{});
    if (N <= WIDE_THREADS * 32) return f(std::integral_constant<int, 32>{});
    return BG_ERR_UNSUPPORTED_N;
}

template <typename F>
int dispatch_r(int N, F&& f)
{
    switch (rows_per_lane(N)) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 5: return f(std::integral_constant<int, 5>{});
        case 6: return f(std::integral_constant<int, 6>{});
        case 8: return f(std::integral_constant<int, 8>{});
        case 9: return f(std::integral_constant<int, 9>{});
        case 10: return f(std::integral_constant<int, 10>{});
        case 12: return f(std::integral_constant<int, 12>{});
        case 16: return f(std::integral_constant<int, 16>{});
        case 24: return f(std::integral_constant<int, 24>{});
        default: return BG_ERR_UNSUPPORTED_N;
    }
}

}  // namespace

extern "C" {

int bg_abi_version(void) { return BG_ABI_VERSION; }

int bg_last_hip_error(void) { return bg::tls_last_hip_error; }

int bg_fom_max_n(void) { return kWideMaxN; }

const char* bg_strerror(int code)
{
    switch (code) {
        case BG_OK: return "ok";
        case BG_ERR_BAD_ARG: return "bad argument";
        case BG_ERR_UNSUPPORTED_N: return "N not supported by the fused kernels (2 <= N <= 8192)";
        case BG_ERR_NONUNIFORM: return "mesh is not uniform";
        case BG_ERR_LAUNCH: return "kernel launch failed (see bg_last_hip_error)";
        case BG_ERR_UNSUPPORTED_R: return "reduced dimension not supported";
        case BG_ERR_PROJECTION: return "unknown projection";
        case BG_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

int bg_fom_run(int N, int B, int nsteps, const double* x, const double* u0, const double* mu1,
               const double* mu2, double dt, double E, double tol, int max_it, int supg, double* hist,
               int32_t* iters, int32_t* flags, void* stream)
{
    if (N < 2 || B < 0 || nsteps < 0 || max_it < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!x || !u0 || !mu1 || !mu2 || !hist || !flags || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    FomArgs a;
    a.x = x; a.u0 = u0; a.mu1 = mu1; a.mu2 = mu2; a.hist = hist; a.iters = iters; a.flags = flags; a.errs = nullptr;
    a.dt = dt; a.E = E; a.tol2 = tol * tol;
    a.N = N; a.B = B; a.nsteps = nsteps; a.max_it = max_it; a.supg = supg & BG_OPT_SUPG;
    const bool nonuniform = (supg & BG_OPT_NONUNIFORM) != 0;
    const dim3 grid((B + WAVES_PER_WG - 1) / WAVES_PER_WG), block(64 * WAVES_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    // one workgroup per sample: beyond one wavefront's registers -- and, on request (BG_OPT_FOM_WIDE, uniform mesh, N > 64),
    // below that too.  It is NOT the default there: measured at N = 1024 (round 3, B = 128 = one sample per second CU)
    // 2.18 us per Picard iteration against 1.86 us for one wavefront per sample -- 4 rows per thread instead of 16 shorten
    // the per-thread work threefold, but the seven workgroup barriers and LDS exchanges of an iteration cost more than that.
    const bool both = N <= kWaveMaxN && N > 64 && !nonuniform;
    const bool wide = N > kWaveMaxN || (both && (supg & BG_OPT_FOM_WIDE) && !(supg & BG_OPT_FOM_WAVE));
    if (wide)
        return dispatch_wide(N, [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            if constexpr (R >= 8) {
                if (nonuniform) {
                    hipLaunchKernelGGL((fom_wide_kernel<R, false>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
                    return check_launch();
                }
            }
            hipLaunchKernelGGL((fom_wide_kernel<R, true>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
            return check_launch();
        });
    return dispatch_r(N, [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if (nonuniform)
            hipLaunchKernelGGL((fom_fused_kernel<R, false, false>), grid, block, 0, st, a);
        else if (N == 64 * R)
            hipLaunchKernelGGL((fom_fused_kernel<R, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((fom_fused_kernel<R, false>), grid, block, 0, st, a);
        return check_launch();
    });
}

int bg_fom_run_traced(int N, int B, int nsteps, const double* x, const double* u0, const double* mu1, const double* mu2,
                      double dt, double E, double tol, int max_it, int supg, double* hist, int32_t* iters, int32_t* flags,
                      double* errs, void* stream)
{
    if (N < 2 || B < 0 || nsteps < 0 || max_it < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (N > kWaveMaxN) return BG_ERR_UNSUPPORTED_N;         // the wave-per-sample kernels only
    if (B == 0) return BG_OK;
    if (!x || !u0 || !mu1 || !mu2 || !hist || !flags || !errs || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    FomArgs a;
    a.x = x; a.u0 = u0; a.mu1 = mu1; a.mu2 = mu2; a.hist = hist; a.iters = iters; a.flags = flags; a.errs = errs;
    a.dt = dt; a.E = E; a.tol2 = tol * tol;
    a.N = N; a.B = B; a.nsteps = nsteps; a.max_it = max_it; a.supg = supg & BG_OPT_SUPG;
    const bool nonuniform = (supg & BG_OPT_NONUNIFORM) != 0;
    const dim3 grid((B + WAVES_PER_WG - 1) / WAVES_PER_WG), block(64 * WAVES_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    return dispatch_r(N, [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if (nonuniform)
            hipLaunchKernelGGL((fom_fused_kernel<R, false, false, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((fom_fused_kernel<R, false, true, true>), grid, block, 0, st, a);
        return check_launch();
    });
}

int bg_fom_assemble(int N, int B, const double* x, const double* uk, const double* un, const double* mu1,
                    const double* mu2, double dt, double E, int supg, double* lo, double* di, double* up,
                    double* rhs, void* stream)
{
    if (N < 2 || B < 0 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!x || !uk || !un || !mu1 || !mu2 || !lo || !di || !up || !rhs) return BG_ERR_BAD_ARG;
    AsmArgs a;
    a.x = x; a.uk = uk; a.un = un; a.mu1 = mu1; a.mu2 = mu2; a.lo = lo; a.di = di; a.up = up; a.rhs = rhs;
    a.dt = dt; a.E = E; a.N = N; a.B = B; a.supg = supg & BG_OPT_SUPG;
    const bool nonuniform = (supg & BG_OPT_NONUNIFORM) != 0;
    const dim3 grid((B + WAVES_PER_WG - 1) / WAVES_PER_WG), block(64 * WAVES_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    if (N > kWaveMaxN)
        return dispatch_wide(N, [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            if (nonuniform)
                hipLaunchKernelGGL((fom_assemble_wide_kernel<R, false>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
            else
                hipLaunchKernelGGL((fom_assemble_wide_kernel<R, true>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
            return check_launch();
        });
    return dispatch_r(N, [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if (nonuniform)
            hipLaunchKernelGGL((fom_assemble_kernel<R, false, false>), grid, block, 0, st, a);
        else if (N == 64 * R)
            hipLaunchKernelGGL((fom_assemble_kernel<R, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((fom_assemble_kernel<R, false>), grid, block, 0, st, a);
        return check_launch();
    });
}

int bg_tridiag_solve(int N, int B, const double* lo, const double* di, const double* up, const double* rhs,
                     double* sol, void* stream)
{
    if (N < 1 || B < 0) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!lo || !di || !up || !rhs || !sol) return BG_ERR_BAD_ARG;
    SolveArgs a{lo, di, up, rhs, sol, N, B};
    const dim3 grid((B + WAVES_PER_WG - 1) / WAVES_PER_WG), block(64 * WAVES_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    if (N > kWaveMaxN)
        return dispatch_wide(N, [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            hipLaunchKernelGGL((tridiag_solve_wide_kernel<R>), dim3(B), dim3(WIDE_THREADS), 0, st, a);
            return check_launch();
        });
    return dispatch_r(N, [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        hipLaunchKernelGGL((tridiag_solve_kernel<R>), grid, block, 0, st, a);
        return check_launch();
    });
}

int bg_transpose_batched(int B, int rows, int cols, const double* in, double* out, void* stream)
{
    if (B < 0 || rows < 0 || cols < 0) return BG_ERR_BAD_ARG;
    if (B == 0 || rows == 0 || cols == 0) return BG_OK;
    if (!in || !out) return BG_ERR_BAD_ARG;
    const dim3 grid((cols + 31) / 32, (rows + 31) / 32, B < 65535 ? B : 65535), block(256);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, (hipStream_t)stream, in, out, B, rows, cols);
    return check_launch();
}

}  // extern "C"
