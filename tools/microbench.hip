// microbench.hip -- per-instruction issue cost with ONE wave per SIMD (the regime of the fused
// FOM kernel at B = 1024) and accuracy of v_rcp_f64 with 0/1/2 Newton steps.
// build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 256;
constexpr int UNR = 16;   // independent chains

template <int OP>
__global__ __launch_bounds__(256, 1) void bench(double* out, long long* cyc, double seed)
{
    double v[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) v[j] = seed + 0.001 * j + 1e-3 * (threadIdx.x & 63);
    const int lane = threadIdx.x & 63;
    const int perm = ((lane + 1) & 63) << 2;
    __shared__ double lds_all[4][128];
    __shared__ double2 lds2_all[4][128];
    double* lds = lds_all[threadIdx.x >> 6];
    double2* lds2 = lds2_all[threadIdx.x >> 6];
    const bool sel = (lane * 2654435761u) & 64;
    const int dynlane = __builtin_amdgcn_readfirstlane((int)(seed * 4.0) + (int)(blockIdx.x & 7));   // wave-uniform, not a constant
    double w2[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) w2[j] = seed * 1.5 + j;
    lds[lane] = 0; lds[lane + 64] = 0; lds2[lane] = make_double2(0, 0); lds2[lane + 64] = make_double2(0, 0);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (OP == 0) v[j] = __builtin_fma(v[j], 1.0000001, 1e-9);
            if (OP == 1) v[j] = v[j] * 1.0000001;
            if (OP == 2) v[j] = v[j] + 1e-9;
            if (OP == 3) v[j] = __builtin_amdgcn_rcp(v[j]);
            if (OP == 4) {   // fp64 move by two DPP movs (wave_shr:1)
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[j]), 0x138, 0xF, 0xF, true);
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[j]), 0x138, 0xF, 0xF, true);
                v[j] = __hiloint2double(hi, lo);
            }
            if (OP == 5) {   // fp64 move by two ds_bpermute
                int lo = __builtin_amdgcn_ds_bpermute(perm, __double2loint(v[j]));
                int hi = __builtin_amdgcn_ds_bpermute(perm, __double2hiint(v[j]));
                v[j] = __hiloint2double(hi, lo);
            }
            if (OP == 6) v[j] = fmax(v[j], 1.0 + 1e-9 * j);
            if (OP == 7) v[j] = (lane & 1) ? v[j] : v[(j + 1) % UNR];   // 2 cndmask
            if (OP == 8) {   // readlane pair + use
                int lo = __builtin_amdgcn_readlane(__double2loint(v[j]), 5);
                int hi = __builtin_amdgcn_readlane(__double2hiint(v[j]), 5);
                v[j] = __builtin_fma(v[j], 0.5, __hiloint2double(hi, lo));
            }
            if (OP == 9) v[j] = sqrt(v[j]);
            if (OP == 10) v[j] = 1.0 / v[j];
            if (OP == 11) {  // float rcp seed path
                float f = (float)v[j];
                v[j] = (double)__builtin_amdgcn_rcpf(f);
            }
            if (OP == 12) {  // LDS round trip: write b64, read2 b64 of neighbours
                lds[lane + 32] = v[j];
                double a = lds[lane + 31], b = lds[lane + 33];
                v[j] = a + b;
            }
            if (OP == 13) {  // row_shr:1 DPP pair
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[j]), 0x111, 0xF, 0xF, true);
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[j]), 0x111, 0xF, 0xF, true);
                v[j] = __hiloint2double(hi, lo);
            }
            if (OP == 14) {  // ds_swizzle pair (swap 16-lane halves within 32: xor 0x10)
                int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v[j]), 0x401F);
                int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v[j]), 0x401F);
                v[j] = __hiloint2double(hi, lo);
            }
            if (OP == 15) {  // 2 readlanes only (consumed by s_add into vgpr via v_mov)
                int lo = __builtin_amdgcn_readlane(__double2loint(v[j]), 5);
                int hi = __builtin_amdgcn_readlane(__double2hiint(v[j]), 7);
                v[j] = __hiloint2double(hi ^ lane, lo);
            }
            if (OP == 16) {  // true select between two registers
                v[j] = sel ? v[j] : w2[j];
            }
            if (OP == 18) v[j] = v[j] + w2[j];                       // add, register operands
            if (OP == 19) v[j] = v[j] * w2[j];                       // mul, register operands
            if (OP == 20) v[j] = __builtin_fma(v[j], 1.0, w2[j]);    // add expressed as fma
            if (OP == 21) v[j] = __builtin_fma(v[j], w2[j], 0.0);    // mul expressed as fma
            if (OP == 22) v[j] = __builtin_fma(v[j], w2[j], w2[(j + 1) % UNR]);  // fma, register operands
            if (OP == 23 || OP == 24) {  // LU-style staged broadcast: 8 readlanes (dynamic / first lane), then 4 FMAs
                if ((j & 3) == 0) {
                    double pv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int lo, hi;
                        if (OP == 23) {
                            lo = __builtin_amdgcn_readlane(__double2loint(v[j + u]), dynlane);
                            hi = __builtin_amdgcn_readlane(__double2hiint(v[j + u]), dynlane);
                        } else {
                            lo = __builtin_amdgcn_readfirstlane(__double2loint(v[j + u]));
                            hi = __builtin_amdgcn_readfirstlane(__double2hiint(v[j + u]));
                        }
                        pv[u] = __hiloint2double(hi, lo);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[j + u] = __builtin_fma(w2[j + u], pv[u], v[j + u] * 0.5);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (OP == 25) {  // LDS broadcast: every lane reads the same 16 bytes, 2 FMAs consume them
                if ((j & 1) == 0) {
                    double2 a = lds2[(it + j) & 127];
                    v[j] = __builtin_fma(w2[j], a.x, v[j] * 0.5);
                    v[j + 1] = __builtin_fma(w2[j + 1], a.y, v[j + 1] * 0.5);
                }
            }
            if (OP == 26) {  // one lane publishes 2 doubles, all lanes read them back (pivot-row round trip)
                if ((j & 1) == 0) {
                    if (lane == dynlane) lds2[j] = make_double2(v[j], v[j + 1]);
                    double2 a = lds2[j];
                    v[j] = __builtin_fma(w2[j], a.x, v[j] * 0.5);
                    v[j + 1] = __builtin_fma(w2[j + 1], a.y, v[j + 1] * 0.5);
                }
            }
            if (OP == 17) {  // LDS: one b128 write (2 doubles) + one b128 read at neighbour
                lds2[lane + 32] = make_double2(v[j], v[j]);
                double2 a = lds2[lane + 31];
                v[j] = a.x + a.y;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int j = 0; j < UNR; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// dependent chain latency
template <int OP>
__global__ __launch_bounds__(256, 1) void chain(double* out, long long* cyc, double seed)
{
    double v = seed + 1e-3 * (threadIdx.x & 63);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (OP == 0) v = __builtin_fma(v, 1.0000001, 1e-9);
            if (OP == 3) v = __builtin_amdgcn_rcp(v);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void rcp_accuracy(const double* x, double* r0, double* r1, double* r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    r1[i] = r;
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    r2[i] = r;
}

template <int OP>
int run(const char* name, double* out, long long* cyc, int per_iter_instr, int grid = 256)
{
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.5);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.5);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += c; avg /= grid;
    printf("%-28s %8.2f cycles per op-group (%d instr each) -> %.2f cyc/instr  [%d waves/SIMD]\n", name, avg / (ITER * UNR),
           per_iter_instr, avg / (ITER * UNR) / per_iter_instr, grid / 256);
    return 0;
}

int main()
{
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, 1024 * 256 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 1024 * sizeof(long long)));
    printf("grid 256 WG x 256 threads (1 wave/SIMD on every CU), s_memtime-class counter (100 MHz?) see scale below\n");
    run<0>("v_fma_f64", out, cyc, 1);
    run<1>("v_mul_f64", out, cyc, 1);
    run<2>("v_add_f64", out, cyc, 1);
    run<3>("v_rcp_f64", out, cyc, 1);
    run<4>("2x v_mov_b32_dpp", out, cyc, 2);
    run<5>("2x ds_bpermute_b32", out, cyc, 2);
    run<6>("v_max_f64", out, cyc, 1);
    run<7>("2x v_cndmask_b32", out, cyc, 2);
    run<8>("2x v_readlane + fma", out, cyc, 3);
    run<9>("sqrt(f64) libm", out, cyc, 1);
    run<10>("1.0/x f64 (IEEE div)", out, cyc, 1);
    run<11>("cvt+rcp_f32+cvt", out, cyc, 3);
    run<12>("LDS wr b64 + 2 rd b64 + add", out, cyc, 4);
    run<13>("2x v_mov_dpp row_shr:1", out, cyc, 2);
    run<14>("2x ds_swizzle", out, cyc, 2);
    run<15>("2x v_readlane (+xor)", out, cyc, 3);
    run<16>("f64 select (2 cndmask)", out, cyc, 2);
    run<17>("LDS wr b128 + rd b128 + add", out, cyc, 3);
    run<18>("v_add_f64 reg,reg", out, cyc, 1);
    run<19>("v_mul_f64 reg,reg", out, cyc, 1);
    run<20>("fma(x,1.0,y)", out, cyc, 1);
    run<21>("fma(x,y,0.0)", out, cyc, 1);
    run<22>("v_fma_f64 reg,reg,reg", out, cyc, 1);
    // LU broadcast candidates; per op-group = one matrix column (UNR columns per pass)
    for (int g : {256, 1024}) {
        run<22>("fma reg,reg,reg", out, cyc, 1, g);
        run<23>("col: 2 readlane(dyn)+mul+fma", out, cyc, 4, g);
        run<24>("col: 2 readfirstlane+mul+fma", out, cyc, 4, g);
        run<25>("col: 1/2 ds_read_b128 bcast+mul+fma", out, cyc, 3, g);
        run<26>("col: 1/2 (1-lane wr + rd b128)+mul+fma", out, cyc, 3, g);
    }
    {
        hipLaunchKernelGGL(chain<0>, dim3(256), dim3(256), 0, 0, out, cyc, 1.5);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(256);
        CHECK(hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0; for (auto c : h) avg += c; avg /= 256;
        printf("dependent v_fma_f64 chain: %.2f per op\n", avg / (ITER * UNR));
        hipLaunchKernelGGL(chain<3>, dim3(256), dim3(256), 0, 0, out, cyc, 1.5);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
        avg = 0; for (auto c : h) avg += c; avg /= 256;
        printf("dependent v_rcp_f64 chain: %.2f per op\n", avg / (ITER * UNR));
    }
    // counter scale: time a known-duration kernel
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), 0, 0, out, cyc, 1.5);
        hipEventRecord(e1); CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256);
        CHECK(hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0; for (auto c : h) avg += c; avg /= 256;
        printf("fma kernel: %.1f us wall, %.0f counter ticks -> %.1f ticks/us\n", ms * 1e3, avg, avg / (ms * 1e3));
    }
    // rcp accuracy
    {
        const int n = 1 << 20;
        std::vector<double> x(n), a(n), b(n), c(n);
        unsigned long long s = 88172645463325252ull;
        for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = 1e-3 + (s % 1000000007ull) / 1000000007.0 * 10.0; }
        double *dx, *d0, *d1, *d2;
        CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&d0, n * 8)); CHECK(hipMalloc(&d1, n * 8)); CHECK(hipMalloc(&d2, n * 8));
        CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rcp_accuracy, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
        CHECK(hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost));
        double e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < n; ++i) {
            long double t = 1.0L / (long double)x[i];
            e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
            e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
            e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
        }
        printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e (eps = 1.11e-16)\n", e0, e1, e2);
    }
    return 0;
}
