// svd.hip -- one-sided (Hestenes) Jacobi sweep for the small core of the snapshot SVD.
//
// The offline POD basis (reference: np.linalg.svd of the snapshot matrix, POD/pod.py:84, and of the
// quadratic-manifold snapshots, Quadratic_manifold/build_quadratic_manifold.py:29) is built QR-first on the
// device (burgers_hip/pod.py: S^T = Q R), which leaves the SVD of the N x N triangular core R.  rocSOLVER's
// SVD runs a Jacobi eigensolver on the Gram matrix and loses the small singular triplets (absolute accuracy
// 1e-9 sigma_max measured, tools/time_pod.py); one-sided Jacobi orthogonalises the vectors themselves by
// plane rotations and delivers every singular value to high relative accuracy.
//
// Layout: G [m][ld] row-major, the m ROWS are the vectors being orthogonalised (rows of R = columns of R^T);
// J [m][ld] receives the same rotations starting from the identity.  One launch = one step of a round-robin
// ordering: m/2 disjoint row pairs, one 256-thread workgroup per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "wave_ops.hpp"

namespace {

using namespace bg;

__global__ __launch_bounds__(256) void jacobi_pair_kernel(double* __restrict__ G, double* __restrict__ J, int m, int ld,
                                                          const int32_t* __restrict__ pairs, double tol,
                                                          int32_t* __restrict__ rotations)
{
    __shared__ double s_part[3][4];
    const int p = pairs[2 * blockIdx.x], q = pairs[2 * blockIdx.x + 1];
    if (p < 0 || q < 0 || p >= m || q >= m) return;                 // bye of an odd-sized tournament
    double* gp = G + (size_t)p * ld;
    double* gq = G + (size_t)q * ld;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double x = gp[i], y = gq[i];
        a = __builtin_fma(x, x, a);
        b = __builtin_fma(y, y, b);
        c = __builtin_fma(x, y, c);
    }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if (lane == 0) { s_part[0][w] = a; s_part[1][w] = b; s_part[2][w] = c; }
    __syncthreads();
    const double alpha = (s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3]);
    const double beta = (s_part[1][0] + s_part[1][1]) + (s_part[1][2] + s_part[1][3]);
    const double gamma = (s_part[2][0] + s_part[2][1]) + (s_part[2][2] + s_part[2][3]);
    if (!(fabs(gamma) > tol * sqrt(alpha * beta))) return;          // already orthogonal (or a zero / NaN row)
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
    double* jp = J + (size_t)p * ld;
    double* jq = J + (size_t)q * ld;
    for (int i = tid; i < m; i += 256) {
        const double x = gp[i], y = gq[i];
        gp[i] = cs * x - sn * y;
        gq[i] = sn * x + cs * y;
        const double u = jp[i], v = jq[i];
        jp[i] = cs * u - sn * v;
        jq[i] = sn * u + cs * v;
    }
    if (tid == 0) atomicAdd(rotations, 1);
}

}  // namespace

extern "C" int bg_jacobi_sweep(int m, int ld, double* G, double* J, const int32_t* pairs, int n_steps, int n_pairs,
                               double tol, int32_t* rotations, void* stream)
{
    if (m < 1 || ld < m || n_steps < 0 || n_pairs < 0 || !(tol >= 0.0)) return BG_ERR_BAD_ARG;
    if (n_steps == 0 || n_pairs == 0) return BG_OK;
    if (!G || !J || !pairs || !rotations) return BG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < n_steps; ++s)
        hipLaunchKernelGGL(jacobi_pair_kernel, dim3(n_pairs), dim3(256), 0, st, G, J, m, ld,
                           pairs + (size_t)s * n_pairs * 2, tol, rotations);
    return bg::check_launch();
}
