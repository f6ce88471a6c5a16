// mlp.hip -- the activation stage of a forward-mode MLP evaluation (value and input-Jacobian together).
//
// POD-ANN needs N(q) and dN/dq of a small MLP for every sample and iteration (reference:
// compute_ann_jacobian, FEM/fem_burgers.py:1254-1275, per-sample torch autograd in float32; the model call
// at :1241).  With n inputs the value and the n tangent directions travel as the 1 + n rows of one matrix
// per sample, so every linear layer is ONE library GEMM over B*(1+n) rows, and what is left per layer is
// this kernel: add the bias to the value row, apply the activation to it, scale the tangent rows by the
// activation's derivative.  float32 like the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"

namespace {

template <int ACT>
__global__ __launch_bounds__(256) void mlp_act_jvp_kernel(float* __restrict__ z, const float* __restrict__ bias,
                                                          long long total, int n1, int h, float alpha)
{
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long b = idx / h;
        const int j = (int)(idx - b * h);
        float* p = z + (size_t)b * n1 * h + j;
        const float v = p[0] + (bias ? bias[j] : 0.0f);
        float a = v, d = 1.0f;
        if (ACT == BG_ACT_ELU) {
            const float e = alpha * expf(v);
            a = v > 0.0f ? v : e - alpha;
            d = v > 0.0f ? 1.0f : e;
        } else if (ACT == BG_ACT_RELU) {
            a = v > 0.0f ? v : 0.0f;
            d = v > 0.0f ? 1.0f : 0.0f;
        } else if (ACT == BG_ACT_TANH) {
            a = tanhf(v);
            d = 1.0f - a * a;
        }
        p[0] = a;
        if (ACT != BG_ACT_NONE)
            for (int k = 1; k < n1; ++k) p[(size_t)k * h] *= d;
    }
}

}  // namespace

extern "C" int bg_mlp_act_jvp(int B, int n1, int h, float* z, const float* bias, int act, float alpha, void* stream)
{
    if (B < 0 || n1 < 1 || h < 1) return BG_ERR_BAD_ARG;
    if (act != BG_ACT_NONE && act != BG_ACT_ELU && act != BG_ACT_RELU && act != BG_ACT_TANH) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!z) return BG_ERR_BAD_ARG;
    const long long total = (long long)B * h;
    const long long want = (total + 255) / 256;
    const dim3 grid((unsigned)(want < 65536 ? want : 65536)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (act) {
        case BG_ACT_NONE: hipLaunchKernelGGL(mlp_act_jvp_kernel<BG_ACT_NONE>, grid, block, 0, st, z, bias, total, n1, h, alpha); break;
        case BG_ACT_ELU: hipLaunchKernelGGL(mlp_act_jvp_kernel<BG_ACT_ELU>, grid, block, 0, st, z, bias, total, n1, h, alpha); break;
        case BG_ACT_RELU: hipLaunchKernelGGL(mlp_act_jvp_kernel<BG_ACT_RELU>, grid, block, 0, st, z, bias, total, n1, h, alpha); break;
        default: hipLaunchKernelGGL(mlp_act_jvp_kernel<BG_ACT_TANH>, grid, block, 0, st, z, bias, total, n1, h, alpha); break;
    }
    return bg::check_launch();
}
