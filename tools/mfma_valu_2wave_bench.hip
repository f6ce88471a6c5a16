// mfma_valu_2wave_bench.hip -- do the fp64 matrix instruction of ONE wave and vector-ALU work of ANOTHER wave on the
// same SIMD overlap on gfx950?  (tools/mfma_valu_bench.hip answered the same-wave question: they serialise.)
// 512-thread workgroups, one per CU: waves w and w + 4 share a SIMD.  Waves 0-3 issue 400 x 100 v_mfma_f64_4x4x4_4b
// (independent accumulators, random data), waves 4-7 issue NV vector instructions of one kind (8 independent chains).
// Three launches per kind: matrix waves alone, vector waves alone, both.  both ~ max(alone) = the pipes overlap across
// waves; both ~ sum = they share one datapath.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>      // 0: v_fma_f64, 1: v_mov_b32 dpp, 2: v_fma_f32, 3: ds_read_b64, 4: v_readlane pairs
__global__ __launch_bounds__(512, 2) void k(double* out, const double* in, int mfma_groups, int valu_iters)
{
    __shared__ double lds[512];
    const int w = threadIdx.x >> 6;
    lds[threadIdx.x] = in[threadIdx.x % 256];
    __syncthreads();
    double s = 0;
    if (w < 4) {
        double acc[100];
        for (int i = 0; i < 100; ++i) acc[i] = 0.0;
        double A[10], Bv[10];
        for (int i = 0; i < 10; ++i) A[i] = in[(threadIdx.x & 255) * 21 + i];
        for (int i = 0; i < 10; ++i) Bv[i] = in[(threadIdx.x & 255) * 21 + 10 + i];
        for (int it = 0; it < mfma_groups; ++it) {
            int p = 0;
#pragma unroll
            for (int a = 0; a < 10; ++a)
#pragma unroll
                for (int b = 0; b < 10; ++b, ++p) acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[a], Bv[b], acc[p], 0, 0, 0);
        }
        for (int i = 0; i < 100; ++i) s += acc[i];
    } else {
        double x[8]; float xf[8];
        for (int i = 0; i < 8; ++i) { x[i] = in[(threadIdx.x & 255) * 21 + i] * 0.5; xf[i] = (float)x[i]; }
        const double c1 = in[0] * 1e-3 + 0.999, c2 = in[1] * 1e-3;
        int addr = (threadIdx.x & 63) * 8;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const int v = q & 7;
                if (KIND == 0) x[v] = __builtin_fma(x[v], c1, c2);
                else if (KIND == 2) xf[v] = __builtin_fmaf(xf[v], (float)c1, (float)c2);
                else if (KIND == 1) { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[v]), 0x111, 0xF, 0xF, true); x[v] = __hiloint2double(__double2hiint(x[v]), lo); }
                else if (KIND == 3) { x[v] += lds[(addr / 8 + q) & 511]; }
                else { x[v] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[v]), q & 63), __builtin_amdgcn_readlane(__double2loint(x[v]), q & 63)); }
            }
        }
        for (int i = 0; i < 8; ++i) s += x[i] + xf[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND> float launch(double* out, const double* in, int mg, int vi)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, in, mg, vi);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, in, mg, vi);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
template <int KIND> void run(double* out, const double* in, const char* what, int vi)
{
    const int mg = 400;
    const float tm = launch<KIND>(out, in, mg, 0), tv = launch<KIND>(out, in, 0, vi), tb = launch<KIND>(out, in, mg, vi);
    printf("%-12s matrix waves alone %7.1f us (%5.2f ns/MFMA) | vector waves alone %7.1f us (%5.2f ns/instr) | both %7.1f us "
           "| max %7.1f  sum %7.1f -> overlap fraction %.2f\n", what, tm, tm * 1e3 / (mg * 100), tv, tv * 1e3 / (vi * 64.0), tb,
           tm > tv ? tm : tv, tm + tv, (tm + tv - tb) / (tm < tv ? tm : tv));
}
int main()
{
    double *out, *in;
    (void)hipMalloc(&out, 256 * 512 * 8); (void)hipMalloc(&in, 256 * 21 * 8);
    std::vector<double> h(256 * 21);
    unsigned long long sd = 88172645463325252ull;
    for (auto& v : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = (double)(sd % 2000003) / 1000001.0 - 1.0; }
    (void)hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    run<0>(out, in, "v_fma_f64", 2400); run<1>(out, in, "v_mov dpp", 2400); run<2>(out, in, "v_fma_f32", 2400);
    run<3>(out, in, "ds_read_b64", 1200); run<4>(out, in, "v_readlane", 1200);
    return 0;
}
