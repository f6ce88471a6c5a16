// Store-pattern microbenchmark for the decoder's result layout out[B][N][Nt] (float64, Nt = 501: row pitch 4008 bytes):
// pure stores, no arithmetic, 2.1 GB per launch.  Patterns (one 256-thread workgroup each, two per CU by LDS padding off):
//   0  decoder tile: 128 consecutive (sample, time) columns x all 512 rows -- 512 runs of 1 KB, pitch 4008 B, 16 B per lane
//   1  contiguous: every workgroup writes 512 KB in one piece (fill_ order)
//   2  row block: 16 rows x all 501 levels of one sample = 64 KB contiguous per wave of 4 rows
//   3  decoder tile with 8-byte stores (two per lane)
//   4  one workgroup per sample, wave w writes the columns 128 w .. + 127 of every row, the four waves row by row together
//   5  decoder tile, XCD-aware order: workgroup i works on chunk (i % 8) * (chunks / 8) + i / 8, so that the neighbouring
//      chunks of a sample run at the same time on the SAME XCD (one L2 sees the whole 4-KB row)
//   6  as 5 with tiles of 256 columns (2-KB runs, each wave two 1-KB stores per row)
// build: hipcc --offload-arch=gfx950 -O3 tools/store_pattern_bench.hip -o build/store_pattern_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int N = 512, Nt = 501;
template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(double* __restrict__ out, long long C, int B)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double v = (double)blockIdx.x;
    if (PAT == 0 || PAT == 3 || PAT == 5) {
        long long chunk = blockIdx.x;
        if (PAT == 5) {
            const long long per = (gridDim.x + 7) / 8;
            chunk = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
            if (chunk * 128 >= C) return;
        }
        const long long c0 = chunk * 128;
        const long long ca = c0 + 2 * lane, cb = ca + 1;
        const long long ba = (ca < C ? ca : 0) / Nt, bb = (cb < C ? cb : 0) / Nt;
        const long long o0 = ca < C ? ba * (long long)N * Nt + (ca - ba * Nt) : -1, o1 = cb < C ? bb * (long long)N * Nt + (cb - bb * Nt) : -1;
        const bool pair = o0 >= 0 && o1 == o0 + 1;
        for (int row = w; row < N; row += 4) {
            const long long ro = (long long)row * Nt;
            if (PAT != 3 && pair) *reinterpret_cast<double2*>(out + o0 + ro) = make_double2(v, v);
            else { if (o0 >= 0) out[o0 + ro] = v; if (o1 >= 0) out[o1 + ro] = v; }
        }
    } else if (PAT == 6) {
        const long long per = (gridDim.x + 7) / 8;
        const long long chunk = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
        const long long c0 = chunk * 256;
        if (c0 >= C) return;
        long long o[2];
        bool pr[2];
        for (int h = 0; h < 2; ++h) {
            const long long ca = c0 + 128 * h + 2 * lane, cb = ca + 1;
            const long long ba = (ca < C ? ca : 0) / Nt, bb = (cb < C ? cb : 0) / Nt;
            const long long o0 = ca < C ? ba * (long long)N * Nt + (ca - ba * Nt) : -1, o1 = cb < C ? bb * (long long)N * Nt + (cb - bb * Nt) : -1;
            o[h] = o0; pr[h] = o0 >= 0 && o1 == o0 + 1;
        }
        for (int row = w; row < N; row += 4) {
            const long long ro = (long long)row * Nt;
            for (int h = 0; h < 2; ++h) {
                if (pr[h]) *reinterpret_cast<double2*>(out + o[h] + ro) = make_double2(v, v);
                else if (o[h] >= 0) out[o[h] + ro] = v;
            }
        }
    } else if (PAT == 4) {
        double* p = out + (size_t)blockIdx.x * N * Nt + 128 * w + 2 * lane;               // sample = blockIdx
        const bool in2 = 128 * w + 2 * lane + 1 < Nt, in1 = 128 * w + 2 * lane < Nt;
        for (int row = 0; row < N; ++row) {
            if (in2) { p[(size_t)row * Nt] = v; p[(size_t)row * Nt + 1] = v; }
            else if (in1) p[(size_t)row * Nt] = v;
        }
    } else if (PAT == 1) {
        double2* p = reinterpret_cast<double2*>(out) + (size_t)blockIdx.x * 32768;          // 512 KB per workgroup
        const size_t total = (size_t)B * N * Nt / 2;
        for (int i = tid; i < 32768; i += 256) if ((size_t)blockIdx.x * 32768 + i < total) p[i] = make_double2(v, v);
    } else {
        // blockIdx = (sample, 16-row block): wave w rows 4 w .. 4 w + 3 = 4 x 4008 bytes contiguous
        const int b = blockIdx.x / (N / 16), rb = blockIdx.x % (N / 16);
        double* p = out + ((size_t)b * N + 16 * rb + 4 * w) * Nt;
        for (int i = lane; i < 4 * Nt; i += 64) p[i] = v;
    }
}
int main()
{
    const int B = 1024;
    const long long C = (long long)B * Nt;
    double* out; CK(hipMalloc(&out, (size_t)B * N * Nt * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (double)B * N * Nt * 8 / 1e9;
    auto run = [&](int pat, auto kern, int grid) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, C, B);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, C, B);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("pattern %d: %.3f ms, %.2f TB/s\n", pat, ms, gb / ms);
    };
    run(0, store_kernel<0>, (int)((C + 127) / 128));
    run(1, store_kernel<1>, (int)(((size_t)B * N * Nt / 2 + 32767) / 32768));
    run(2, store_kernel<2>, B * (N / 16));
    run(3, store_kernel<3>, (int)((C + 127) / 128));
    run(4, store_kernel<4>, B);
    run(5, store_kernel<5>, (int)(((C + 127) / 128 + 7) / 8 * 8));
    run(6, store_kernel<6>, (int)(((C + 255) / 256 + 7) / 8 * 8));
    CK(hipMemset(out, 0, 1024));
    return 0;
}
