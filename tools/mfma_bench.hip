// mfma_bench.hip -- issue rate of v_mfma_f64_16x16x4_f64 on gfx950, one wave per SIMD,
// 9 independent accumulators (the shape of the ROM reduce kernel's inner step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f64x4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(256, 1) void k(double* out, long long* cyc, double seed, int iters)
{
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256, 1) void k4(double* out, long long* cyc, double seed, int iters)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// realistic pattern: 10 A operands x 11 B operands -> 110 accumulators, data-dependent values
__global__ __launch_bounds__(256, 1) void k4_pairs(double* out, long long* cyc, const double* in, int iters)
{
    double acc[110];
    for (int i = 0; i < 110; ++i) acc[i] = 0.0;
    double A[10], Bv[11];
    for (int i = 0; i < 10; ++i) A[i] = in[threadIdx.x * 21 + i];
    for (int i = 0; i < 11; ++i) Bv[i] = in[threadIdx.x * 21 + 10 + i];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int p = 0;
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
            for (int b = 0; b < 11; ++b, ++p) acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[a], Bv[b], acc[p], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 10; ++i) A[i] = A[i] * 0.999 + 1e-3;      // keep operands changing (VALU beside MFMA)
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 110; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
void run_pairs(double* out, long long* cyc)
{
    const int iters = 200;
    double* in; (void)hipMalloc(&in, 256 * 21 * 8);
    std::vector<double> h(256 * 21);
    unsigned long long sd = 88172645463325252ull;
    for (auto& v : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = (double)(sd % 2000003) / 1000001.0 - 1.0; }
    (void)hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k4_pairs, dim3(256), dim3(256), 0, 0, out, cyc, in, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k4_pairs, dim3(256), dim3(256), 0, 0, out, cyc, in, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(256);
    (void)hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : hc) avg += c; avg /= 256;
    double n = (double)iters * 110;
    printf("4x4x4_4b 10x11 pairs, random operands: %.1f ticks per MFMA, %.1f ns per MFMA per wave, %.1f TFLOP/s chip\n", avg / n,
           ms * 1e6 / n, n * 512.0 * 1024.0 / (ms * 1e-3) / 1e12);
}
template <int NACC> void run4(double* out, long long* cyc)
{
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k4<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k4<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    double n = (double)iters * NACC;
    double flops = n * 512.0 * 1024.0;
    printf("4x4x4_4b NACC=%d: %.1f ticks per MFMA, %.1f ns per MFMA per wave, %.1f TFLOP/s chip\n", NACC, avg / n,
           ms * 1e6 / n, flops / (ms * 1e-3) / 1e12);
}
template <int NACC> void run(double* out, long long* cyc)
{
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    double n = (double)iters * NACC;
    double flops = n * 2048.0 * 1024.0;   // 1024 waves
    printf("NACC=%d: %.1f counter ticks per MFMA, %.3f ms wall -> %.1f ns per MFMA per wave, %.1f TFLOP/s chip\n", NACC,
           avg / n, ms, ms * 1e6 / n, flops / (ms * 1e-3) / 1e12);
}
int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 256 * 8);
    run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<9>(out, cyc);
    run4<1>(out, cyc); run4<4>(out, cyc); run4<16>(out, cyc);
    run_pairs(out, cyc);
    return 0;
}
