// mfma_layout_probe.hip -- operand / result lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950,
// found with one-hot operands (exact data, as the programming guide asks before relying on a map).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int la, int lb, double* out)
{
    int l = threadIdx.x;
    double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[l] = d;
}
int main()
{
    double* out; (void)hipMalloc(&out, 64 * 8);
    std::vector<double> h(64);
    // for every (la, lb): which D lane becomes 1?
    std::vector<int> hit(64 * 64, -1);
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, lb, out);
            (void)hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) if (h[l] != 0.0) hit[la * 64 + lb] = l;
        }
    // hypothesis: lane = 16*blk + 4*x + y.  Print, for la fixed, the lb's that hit and the D lane.
    for (int la = 0; la < 64; la += 1) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (hit[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, hit[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
