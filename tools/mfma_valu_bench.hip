// mfma_valu_bench.hip -- does fp64 VALU work hide behind v_mfma_f64_4x4x4_4b_f64 on gfx950?
// One wave per SIMD, 110 independent accumulators (10 A x 11 B operands, random data), V independent fp64 FMAs
// interleaved per 110 MFMAs, V = 0 .. 330.  Prints ns per 110-MFMA group: flat in V = the VALU hides, rising = it adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V, int KIND>      // KIND 0: v_fma_f64, 1: v_mov_b32 (dpp-like 32-bit moves), 2: f32 FMA
__global__ __launch_bounds__(256, 1) void k(double* out, const double* in, int iters)
{
    double acc[110];
    for (int i = 0; i < 110; ++i) acc[i] = 0.0;
    double A[10], Bv[11];
    for (int i = 0; i < 10; ++i) A[i] = in[threadIdx.x * 21 + i];
    for (int i = 0; i < 11; ++i) Bv[i] = in[threadIdx.x * 21 + 10 + i];
    constexpr int NV = V > 0 ? (V < 22 ? V : 22) : 1;
    double x[NV]; float xf[NV];
    for (int i = 0; i < NV; ++i) { x[i] = in[threadIdx.x * 21 + (i % 21)] * 0.5; xf[i] = (float)x[i]; }
    const double c1 = in[0] * 1e-3 + 0.999, c2 = in[1] * 1e-3;
    for (int it = 0; it < iters; ++it) {
        int p = 0, v = 0;
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
            for (int b = 0; b < 11; ++b, ++p) {
                acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[a], Bv[b], acc[p], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < (V + 109 - p) / 110; ++q, ++v) {        // V FMAs spread evenly over the 110 MFMAs
                    if (KIND == 0) x[v % NV] = __builtin_fma(x[v % NV], c1, c2);
                    else if (KIND == 2) xf[v % NV] = __builtin_fmaf(xf[v % NV], (float)c1, (float)c2);
                    else { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[v % NV]), 0x111, 0xF, 0xF, true); x[v % NV] = __hiloint2double(__double2hiint(x[v % NV]), lo); }
                }
            }
    }
    double s = 0;
    for (int i = 0; i < 110; ++i) s += acc[i];
    for (int i = 0; i < NV; ++i) s += x[i] + xf[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V, int KIND> void run(double* out, const double* in, const char* what)
{
    const int iters = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, KIND>), dim3(256), dim3(256), 0, 0, out, in, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, KIND>), dim3(256), dim3(256), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s V=%3d per 110 MFMAs: %7.1f ns per group, %5.2f ns per MFMA\n", what, V, ms * 1e6 / iters, ms * 1e6 / iters / 110);
}
int main()
{
    double *out, *in;
    (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&in, 256 * 21 * 8);
    std::vector<double> h(256 * 21);
    unsigned long long sd = 88172645463325252ull;
    for (auto& v : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = (double)(sd % 2000003) / 1000001.0 - 1.0; }
    (void)hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    run<0, 0>(out, in, "fma_f64"); run<55, 0>(out, in, "fma_f64"); run<110, 0>(out, in, "fma_f64"); run<220, 0>(out, in, "fma_f64"); run<330, 0>(out, in, "fma_f64");
    run<110, 1>(out, in, "dpp_mov32"); run<220, 1>(out, in, "dpp_mov32"); run<330, 1>(out, in, "dpp_mov32");
    run<110, 2>(out, in, "fma_f32"); run<220, 2>(out, in, "fma_f32"); run<330, 2>(out, in, "fma_f32");
    return 0;
}
