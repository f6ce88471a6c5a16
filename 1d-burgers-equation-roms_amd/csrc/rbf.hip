// rbf.hip -- kernel values and kernel-gradient factors of the scaled RBF closure, one pass per batch.
//
// POD-RBF (reference: FEM/fem_burgers.py:160-260, used by pod_rbf_prom :1278-1398) evaluates, for every
// sample b and training centre i,
//     xs = 2 (q_p - x_min) / dx - 1,   d_ik = xs_k - Xt_ik,   r2_i = sum_k d_ik^2,
//     gaussian:  phi_i = exp(-eps^2 r2_i),            dphi/dxs_k = -2 eps^2 phi_i   d_ik
//     imq:       phi_i = (1 + eps^2 r2_i)^(-1/2),     dphi/dxs_k =   -eps^2 phi_i^3 d_ik
// and then contracts with the weights: q_s ~ phi W, dq_s/dq_p ~ (dphi/dxs)^T W (library GEMMs).  This
// kernel writes phi [B][Ns] and the gradient factors already transposed and scaled by dxs/dq_p = 2/dx_k,
// GT [B][n][Ns], so that both contractions are ONE GEMM each over B resp. B n rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"

namespace {

template <int KIND>
__global__ __launch_bounds__(256) void rbf_eval_kernel(const double* __restrict__ qp, const double* __restrict__ x_min,
                                                       const double* __restrict__ dx, const double* __restrict__ XtT,
                                                       double* __restrict__ phi, double* __restrict__ GT, int B, int n,
                                                       int Ns, double eps2)
{
    extern __shared__ double s_xs[];                     // [n] scaled coordinates, [n] 2/dx
    double* s_sc = s_xs + n;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += 256) {
            s_xs[k] = 2.0 * ((qp[(size_t)b * n + k] - x_min[k]) / dx[k]) - 1.0;
            s_sc[k] = 2.0 / dx[k];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < Ns; i += 256) {
            double r2 = 0.0;
            for (int k = 0; k < n; ++k) {
                const double d = s_xs[k] - XtT[(size_t)k * Ns + i];
                r2 = __builtin_fma(d, d, r2);
            }
            double p, coef;
            if (KIND == BG_RBF_GAUSSIAN) {
                p = exp(-eps2 * r2);
                coef = -2.0 * eps2 * p;
            } else {
                p = 1.0 / sqrt(1.0 + eps2 * r2);
                coef = -eps2 * (p * p * p);
            }
            if (phi) phi[(size_t)b * Ns + i] = p;
            if (GT) {
                double* g = GT + (size_t)b * n * Ns + i;
                for (int k = 0; k < n; ++k)
                    g[(size_t)k * Ns] = (coef * s_sc[k]) * (s_xs[k] - XtT[(size_t)k * Ns + i]);
            }
        }
    }
}

}  // namespace

extern "C" int bg_rbf_eval(int B, int n, int Ns, int kind, double eps, const double* qp, const double* x_min,
                           const double* dx, const double* XtT, double* phi, double* GT, void* stream)
{
    if (B < 0 || n < 1 || Ns < 1) return BG_ERR_BAD_ARG;
    if (kind != BG_RBF_GAUSSIAN && kind != BG_RBF_IMQ) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!qp || !x_min || !dx || !XtT || (!phi && !GT)) return BG_ERR_BAD_ARG;
    const dim3 grid(B < 65535 ? B : 65535), block(256);
    const size_t lds = 2 * (size_t)n * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
    if (kind == BG_RBF_GAUSSIAN)
        hipLaunchKernelGGL(rbf_eval_kernel<BG_RBF_GAUSSIAN>, grid, block, lds, st, qp, x_min, dx, XtT, phi, GT, B, n, Ns, eps * eps);
    else
        hipLaunchKernelGGL(rbf_eval_kernel<BG_RBF_IMQ>, grid, block, lds, st, qp, x_min, dx, XtT, phi, GT, B, n, Ns, eps * eps);
    return bg::check_launch();
}
