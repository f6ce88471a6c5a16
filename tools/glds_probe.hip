// glds_probe.hip -- semantics of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950:
// per-lane global source, LDS destination = wave-uniform base + lane*16.  Copies 512 doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__global__ __launch_bounds__(256) void k(const double* src, double* dst, int n)
{
    __shared__ __attribute__((aligned(16))) double buf[512];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int idx = w * 128 + lane * 2;                 // 2 doubles per lane
    if (idx < n)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + idx), (lds_void*)(buf + w * 128), 16, 0, 0);
    __syncthreads();
    for (int i = tid; i < 512; i += 256) dst[i] = (i < n) ? buf[i] : -1.0;
}
int main()
{
    const int n = 500;
    std::vector<double> h(512), o(512);
    for (int i = 0; i < 512; ++i) h[i] = i + 0.25;
    double *s, *d; (void)hipMalloc(&s, 512 * 8); (void)hipMalloc(&d, 512 * 8);
    (void)hipMemcpy(s, h.data(), 512 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d, n);
    (void)hipMemcpy(o.data(), d, 512 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; ++i) { double e = i < n ? h[i] : -1.0; if (o[i] != e) { if (bad < 8) printf("i=%d got %g want %g\n", i, o[i], e); ++bad; } }
    printf("glds probe: %d mismatches\n", bad);
    return bad != 0;
}
