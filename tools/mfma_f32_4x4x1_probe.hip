// Operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, read off two products:
//   run 0: a = lane, b = 1  ->  D[lane][reg] = lane that supplied the A row of this element
//   run 1: a = 1, b = lane  ->  D[lane][reg] = lane that supplied the B column of this element
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f32_4x4x1_probe.hip -o build/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out)
{
    const int lane = threadIdx.x;
    for (int run = 0; run < 2; ++run) {
        const float a = run == 0 ? (float)lane : 1.0f, b = run == 0 ? 1.0f : (float)lane;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(run * 64 + lane) * 4 + r] = c[r];
    }
}
int main()
{
    float* d; hipMalloc(&d, 2 * 64 * 4 * sizeof(float));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[2 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int run = 0; run < 2; ++run) {
        printf("run %d (%s supplier lane) per lane: reg0 reg1 reg2 reg3\n", run, run ? "B" : "A");
        for (int l = 0; l < 64; l += 1) if (l < 12 || l >= 60) printf("  lane %2d: %3.0f %3.0f %3.0f %3.0f\n", l, h[(run * 64 + l) * 4], h[(run * 64 + l) * 4 + 1], h[(run * 64 + l) * 4 + 2], h[(run * 64 + l) * 4 + 3]);
    }
    return 0;
}
