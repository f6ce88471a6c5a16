// rom_ann_fused.hip -- the whole POD-ANN PROM time loop of one sample on one compute unit, and its C-ABI entry point.
//
// Replaces FEMBurgers.pod_ann_prom (reference FEM/fem_burgers.py:1177-1251, compute_ann_jacobian :1254-1275) for a batch of
// samples when the closure is a plain MLP (Linear + ELU / ReLU / Tanh stacks, POD-ANN/pod_ann.py:38-56): one 256-thread
// workgroup owns a sample for ALL time steps and Gauss-Newton iterations.  Per iteration, with no kernel boundary:
//     assembly (2 rows per thread)
//     -> projection of the tangent W = U_p + U_s dN on v_mfma_f64_4x4x4_4b (mfma_pass, as bg_rom_run; W lives in LDS)
//     -> n x n solve with partial pivoting (one wave, n <= 8)              np.linalg.solve :1237
//     -> q_p += dq, err = |dq| / (|q_p| + 1e-14), stopping test            :1238-1244
//     -> closure at the new q_p: the MLP value N(q_p) AND its input-Jacobian in ONE forward-mode pass in float32 (the
//        value and the n tangent directions are the 1 + n rows of a small matrix in LDS; a thread owns 4 outputs of a
//        layer and a slice of its inputs, the weights stream from L2 as 16-byte loads of W^T) -- the reference evaluates both in float32 too
//        (:1219, :1241) -- then ONE sweep over U_s^T that forms the decode u = U_p q_p + U_s N(q_p) (:1242) and the next
//        iteration's tangent W = U_p + U_s dN (:1224) together, straight into LDS.
// HBM sees u0 once and one N-row history write per time step; the MLP weights (0.5 MB) and U_s (0.4 MB) are re-read from
// L2 every iteration.  Host-side the batched iteration of burgers_hip/rom.py needs ~40 dependent launches per iteration
// for the same work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "rom_fused_device.hpp"

namespace {

using namespace bg;
using namespace bg::fused;

// Issue priority of the phases (see rom_fused.hip): 3 by default, 0 in the phases named by this bit mask -- 1 the sweep
// over [U_p | U_s], 2 the closure MLP, 4 the projection.  Measured (B = 2048): 0 (no priorities) 1.312e7, 1: 1.322e7,
// 3: 1.329e7, 5 / 7: 1.327e7 -- the solve, assembly and update chains of one workgroup no longer queue behind the other's bulk phases.
#ifndef BG_ANN_PRIO
#define BG_ANN_PRIO 3
#endif
constexpr int ANN_MAX_LAYERS = 8;
constexpr int ANN_MAX_WIDTH = 256;      // one thread per neuron
constexpr int ANN_MAX_N = 8;            // reduced coordinates: two 4-column MFMA blocks
constexpr int ANN_MAX_ROWS = 1 + ANN_MAX_N;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// a + (a of the neighbouring row) in the even rows of 16 lanes, b + (b of the neighbouring row) in the odd rows
__device__ __forceinline__ float swap16_add(float a, float b)
{
    const auto t = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)t[0]) + __builtin_bit_cast(float, (unsigned)t[1]);
}
// a + (a of the other half) in lanes 0..31, b + (b of the other half) in lanes 32..63
__device__ __forceinline__ float swap32_add(float a, float b)
{
    const auto t = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)t[0]) + __builtin_bit_cast(float, (unsigned)t[1]);
}

// value of lane `src` (any lane), two ds_bpermute
__device__ __forceinline__ double lane_f64(double v, int src)
{
    const int lo = __builtin_amdgcn_ds_bpermute(4 * src, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(4 * src, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// x = solve(Ar, -br) for n <= 8 with np.linalg.solve's pivot choice (the not-yet-used row with the largest |a[k]|, ties
// to the lowest row, as lu_pivoted_wave), one wavefront, the WHOLE system spread over the lanes: lane 8 i + j holds
// a[i][j] and a copy of b[i].  Per step: the column entry of the own row and the pivot row's entry of the own column by
// ds_bpermute, one rank-1 update of all 64 entries; the rows above the pivot are eliminated too (Gauss-Jordan), so what is
// left is diagonal and there is no back substitution.  lu_pivoted_wave<8> (lane = row, eight serial steps whatever n is,
// a readlane pair per entry, then eight dependent back-substitution steps) took 10.8 k clocks per call here.
template <bool GAL>
__device__ __forceinline__ void pivoted_gj8(const double (*__restrict__ s_red)[8][12], double* __restrict__ s_x,
                                            int* __restrict__ s_info, int lane, int n)
{
    const int i = lane >> 3, j = lane & 7;
    auto entry = [&](int r, int c) -> double {               // (Ar | br)[r][c], c <= 8; LSPG: mirror the lower blocks
        int rr = r, cc = c;
        if (!GAL && c < 8 && (r >> 2) > (c >> 2)) { rr = c; cc = r; }
        return (s_red[0][rr][cc] + s_red[1][rr][cc]) + (s_red[2][rr][cc] + s_red[3][rr][cc]);
    };
    double A = (i < n && j < n) ? entry(i, j) : (i == j ? 1.0 : 0.0);
    double rb = (i < n) ? -entry(i, 8) : 0.0;
    bool used = false;
    int my_step = -1, info = 0;
    if (lane < 8) s_x[lane] = 0.0;
    for (int k = 0; k < n; ++k) {
        const double ck = lane_f64(A, (lane & ~7) | k);                      // a[i][k]
        const unsigned key = used ? 0u : (((unsigned)__double2hiint(ck) & 0x7fffffffu) + 1u);
        const unsigned best = wave_max_u32(key);
        const unsigned long long m = __ballot(key == best && !used);
        const int p = __builtin_ctzll(m) >> 3;                               // lowest candidate row
        const double piv = readlane_f64(A, 8 * p + k);
        if (piv == 0.0 && info == 0) info = k + 1;
        const double rp = rcp(piv);
        const double prow = lane_f64(A, 8 * p + j);                          // a[p][j]
        const double pb = readlane_f64(rb, 8 * p);
        const double mult = (i == p) ? 0.0 : ck * rp;
        A = __builtin_fma(-mult, prow, A);
        rb = __builtin_fma(-mult, pb, rb);
        if (i == p) { used = true; my_step = k; }
    }
    if (j == my_step) s_x[my_step] = rb * rcp(A);                            // the row that pivoted at step k holds x_k
    if (lane == 0) *s_info = info;
}

struct AnnRunArgs {
    const double* x;        // [N]
    const double* UT;       // [m8][N]: rows 0 .. n-1 = U_p^T, rows n .. n+nbar-1 = U_s^T, zero rows up to m8 = (n + nbar) rounded up to 8
    const double* u0;       // [B][N]
    const double* mu1;      // [B]
    const double* mu2;      // [B]
    double* hist;           // [B][nsteps+1][N]
    int32_t* iters;         // [B][nsteps]
    int32_t* flags;         // [B]
    int32_t* info;          // [B]
    const int32_t* order;   // [B] or null: slot i of the persistent loop works on sample order[i]
    const float* wt[ANN_MAX_LAYERS];     // layer l: W^T, [in4][ld] row-major: width[l] rounded up to 4 rows, width[l+1] to 8 columns, zero fill
    const float* bias[ANN_MAX_LAYERS];   // [width[l+1]] or null
    int width[ANN_MAX_LAYERS + 1];
    int act[ANN_MAX_LAYERS];
    float alpha[ANN_MAX_LAYERS];
    int nl;
    double dt, E, tol;
    int N, B, n, nbar, nsteps, max_it, supg, nonuniform, no_reuse;
};

template <int S, int PROJ>
__global__ __launch_bounds__(256, 2) void rom_ann_fused_kernel(AnnRunArgs a)
{
    constexpr int NB = 2;
    constexpr int NPAD = 64 * S;
    constexpr int RW = 4 * NB;
    constexpr bool GAL = PROJ == BG_PROJ_GALERKIN;
    __shared__ double s_u[NPAD + 4];             // u at offset 2, zero halo on each side
    __shared__ double s_g[NPAD], s_h[NPAD];
    __shared__ __attribute__((aligned(16))) double s_W[NPAD + 2][RW];      // tangent rows, row i at [i + 1]; zero rows around and beyond N
    __shared__ double s_wtu[4][RW];
    __shared__ double s_q[RW], s_x[RW];
    __shared__ float s_qlast[RW];                // float32 input of the latest full closure evaluation (see the step start)
    __shared__ double s_part[4][RW];             // per-wave partial sums of U_p^T u
    __shared__ int s_info;
    // The two halves of an iteration never overlap in time and share one block of LDS (two workgroups per CU need
    // <= 80 KB each):  projection + solve: s_coef, s_halo, s_red   |   closure: s_act, s_dN, s_qs
    constexpr int kCoefB = NPAD * 4 * 8, kHaloB = 2 * NB * 256 * 8, kRedB = 4 * RW * (RW + 4) * 8;
    constexpr int kSweepDepth = 3;                 // register buffers of the sweep (groups of 4 modes in flight + 1); 6 measured slower
    constexpr int kModes = 128 + ANN_MAX_N + 4 * kSweepDepth;   // secondary + primary modes of the sweep + zero modes that round its trip count up
    constexpr int kBW = 12;                        // columns of the sweep's right-hand matrix: n derivatives + 1 coefficient, padded to 3 blocks of 4
    constexpr int kBS = 13;                        // its row stride in doubles (odd: the row-per-lane writes spread over the banks)
    constexpr int kActB = 2 * ANN_MAX_ROWS * ANN_MAX_WIDTH * 4, kDnB = kModes * kBS * 8, kQsB = 0;
    constexpr int kPhaseA = kCoefB + kHaloB + kRedB, kPhaseB = kActB + kDnB + kQsB;
    __shared__ __attribute__((aligned(16))) unsigned char s_shared[kPhaseA > kPhaseB ? kPhaseA : kPhaseB];
    auto& s_coef = *reinterpret_cast<double (*)[NPAD][4]>(s_shared);
    auto& s_halo = *reinterpret_cast<double (*)[2][NB][256]>(s_shared + kCoefB);
    auto& s_red = *reinterpret_cast<double (*)[4][RW][RW + 4]>(s_shared + kCoefB + kHaloB);
    auto& s_act = *reinterpret_cast<float (*)[2][ANN_MAX_ROWS][ANN_MAX_WIDTH]>(s_shared);   // MLP activations: value row + n tangent rows
    auto& s_dN = *reinterpret_cast<double (*)[kModes][kBS]>(s_shared + kActB);              // mode j of the sweep: [d(coefficient j)/dq_p | coefficient j | 0]

    // The thread-index family is re-derived from an opaque copy at the top of every Gauss-Newton pass (see the time loop):
    // per-lane addresses are loop invariants of the whole kernel, and hoisted out of the loops by the dozen they end up in
    // scratch (round 3: the same measure took bg_rom_run's kernels from 160 / 280 to 12 / 152 bytes).
    int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction
    int t = lane & 3, owner = 16 * w + (lane >> 2);
    const int N = a.N, n = a.n, nbar = a.nbar, nr = 1 + a.n, m8 = (a.n + a.nbar + 7) & ~7;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    int rowbase = owner * S;
    auto rederive = [&]() {
        int v = threadIdx.x;
        asm volatile("" : "+v"(v));
        tid = v; lane = v & 63; t = lane & 3; owner = 16 * w + (lane >> 2); rowbase = owner * S;
    };

#ifdef BG_ANN_LDS_PAD                              // timing builds only: extra LDS, so that one workgroup fills a CU
    __shared__ volatile int s_pad[BG_ANN_LDS_PAD / 4];
    if (a.B < 0) s_pad[threadIdx.x] = 1;
#endif
    if (tid < 4) s_u[tid < 2 ? tid : NPAD + tid] = 0.0;
    for (int e = tid; e < (NPAD + 2) * RW; e += 256) (&s_W[0][0])[e] = 0.0;
    if (tid < RW) s_q[tid] = 0.0;

    // ---- N(q_p) and dN/dq_p at q_p = s_q, float32 forward mode: rows 0 = value, 1 .. n = tangent directions --------
    long long cyc[16];                           // timing builds only: shader clocks per phase, see the end of the sample loop
#pragma unroll
    for (int i = 0; i < 16; ++i) cyc[i] = 0;
    long long tick = 0;
    auto lap = [&](int i) {
        if constexpr (kTiming) {
            const long long now = (long long)__builtin_amdgcn_s_memtime();
            cyc[i] += now - tick;
            tick = now;
        }
    };
    // ---- N(q_p) and dN/dq_p at q_p = s_q, float32 forward mode: rows 0 = value, 1 .. n = tangent directions --------
    // Layer l on all 256 threads.  A thread owns 8 outputs (two 16-byte weight loads per input k, no guards: the host
    // pads W^T to [in4][ld]) and every KPw-th group of 4 inputs; lane = (input slice) * P + (output group), so that
    // neighbouring lanes read neighbouring 16-byte chunks of a weight row.  NRT = rows compiled in.
    auto mlp_impl = [&](auto nrt_c) __attribute__((always_inline)) {
        constexpr int NRT = decltype(nrt_c)::value;
        int cur = 0;
        if (tid < RW) {                                              // inputs padded with zeros to a multiple of 4
            s_act[0][0][tid] = tid < n ? (float)s_q[tid] : 0.0f;
            s_qlast[tid] = tid < n ? (float)s_q[tid] : 0.0f;
#pragma unroll
            for (int r = 1; r < ANN_MAX_ROWS; ++r) s_act[0][r][tid] = (r - 1 == tid && tid < n) ? 1.0f : 0.0f;
        }
        __syncthreads();
        lap(14);
        for (int l = 0; l < (skip(128) ? 0 : a.nl); ++l) {          // (128: timing builds only)
            const int in4 = (a.width[l] + 3) & ~3, out = a.width[l + 1], ldw = (out + 7) & ~7, ogn = ldw >> 3;
            int P = 1, pshift = 0;
            while (4 * P < ogn) { P <<= 1; ++pshift; }                // output groups per wave (a power of two, <= 8)
            int KPw = 64 >> pshift;                                   // input slices per wave ...
            while (4 * KPw > in4 && KPw > 1) KPw >>= 1;               // ... at most one per group of 4 inputs (small layers)
            const int span = P * KPw;                                 // lanes of a wave that carry partial sums
            const int kp = lane >> pshift, ogr = w * P + (lane & (P - 1));    // neighbouring lanes: neighbouring 16-byte chunks
            const int og = ogr < ogn ? ogr : ogn - 1;                 // spare lanes redo the last group (no guarded loads)
            // the 8 outputs of a thread are two chunks of 4: [4 og, 4 og + 4) and the same in the second half of the row
            const float* __restrict__ wp = a.wt[l] + 4 * og;
            const int half = 4 * ogn;
            // this thread's bias for the activation stage, fetched now so that its latency hides behind the layer
            const float bias_v = a.bias[l] ? a.bias[l][tid < out ? tid : out - 1] : 0.0f;
            // ... and the two outputs this lane finishes when the layer takes the swap fold below
            const int o0 = ((lane >> 4) < 2 ? 4 * og + 2 * (lane >> 4) : half + 4 * og + 2 * ((lane >> 4) - 2));
            const float bias_0 = a.bias[l] ? a.bias[l][o0 < out ? o0 : out - 1] : 0.0f;
            const float bias_1 = a.bias[l] ? a.bias[l][o0 + 1 < out ? o0 + 1 : out - 1] : 0.0f;
            float acc[NRT][8];
#pragma unroll
            for (int r = 0; r < NRT; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] = 0.0f;
            auto kload = [&](int kb, float4 (&wv)[4][2]) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float4* src = reinterpret_cast<const float4*>(wp + (size_t)(kb + kk) * ldw);
                    if constexpr (skip(1024)) {                          // timing only: no weight traffic
                        wv[kk][0] = make_float4(1.0f * kb, 2.0f, 3.0f, 4.0f); wv[kk][1] = wv[kk][0];
                    } else {
                        wv[kk][0] = src[0];
                        wv[kk][1] = src[ogn];
                    }
                }
            };
            auto kfma = [&](int kb, const float4 (&wv)[4][2]) {
#pragma unroll
                for (int r = 0; r < (skip(512) ? 1 : NRT); ++r) {      // (512: timing only: weight traffic, one row of arithmetic)
                    const float4 xv = *reinterpret_cast<const float4*>(&s_act[cur][r][kb]);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        acc[r][0] = __builtin_fmaf(xs[kk], wv[kk][0].x, acc[r][0]); acc[r][1] = __builtin_fmaf(xs[kk], wv[kk][0].y, acc[r][1]);
                        acc[r][2] = __builtin_fmaf(xs[kk], wv[kk][0].z, acc[r][2]); acc[r][3] = __builtin_fmaf(xs[kk], wv[kk][0].w, acc[r][3]);
                        acc[r][4] = __builtin_fmaf(xs[kk], wv[kk][1].x, acc[r][4]); acc[r][5] = __builtin_fmaf(xs[kk], wv[kk][1].y, acc[r][5]);
                        acc[r][6] = __builtin_fmaf(xs[kk], wv[kk][1].z, acc[r][6]); acc[r][7] = __builtin_fmaf(xs[kk], wv[kk][1].w, acc[r][7]);
                    }
                }
            };
            // Two slices per trip: 16 weight loads in flight.  A rolling ring of fetches that also runs ahead into the next
            // layer was tried: with the 256-register budget of two workgroups per CU it spills, and was slower.
            const int kstride = 4 * KPw;
            int kb = kp < KPw ? 4 * kp : in4;                         // lanes beyond the span hold zeros and stay out of the fold
            if constexpr (NRT <= 6) {
                for (; kb + kstride < in4; kb += 2 * kstride) {
                    float4 wa[4][2], wb[4][2];
                    kload(kb, wa);
                    kload(kb + kstride, wb);
                    kfma(kb, wa);
                    kfma(kb + kstride, wb);
                }
            }
            for (; kb < in4; kb += kstride) {
                float4 wa[4][2];
                kload(kb, wa);
                kfma(kb, wa);
            }
            if constexpr (skip(8192)) lap(3);                          // (8192: timing only: sub-phases of a layer summed over the layers)
            // fold the KPw input slices (lanes P apart): inside a row of 16 lanes with shifts towards the higher lanes, so
            // the row total lands in its last P lanes; across the four rows with ds_bpermute
            auto fold = [&](auto get) {
                if constexpr (skip(2048)) return;                     // timing only
#pragma unroll
                for (int r = 0; r < NRT; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[r][c] += get(acc[r][c]);
            };
            if (P <= 1 && span > 1) fold([](float v) { return dpp_f32<0x111>(v); });      // row_shr:1
            if (P <= 2 && span > 2) fold([](float v) { return dpp_f32<0x112>(v); });      // row_shr:2
            if (P <= 4 && span > 4) fold([](float v) { return dpp_f32<0x114>(v); });      // row_shr:4
            if (span > 8) fold([](float v) { return dpp_f32<0x118>(v); });                // row_shr:8
            const int half_og = half + 4 * og;
            const int kind = a.act[l];
            const float alpha = a.alpha[l];
            // bias, activation, derivative scaling: the arithmetic of bg_mlp_act_jvp (csrc/mlp.hip)
            auto activate = [&](float v, float& av, float& d) {
                av = v; d = 1.0f;
                if (kind == BG_ACT_ELU) {
                    const float e = alpha * expf(v);
                    av = v > 0.0f ? v : e - alpha;
                    d = v > 0.0f ? 1.0f : e;
                } else if (kind == BG_ACT_RELU) {
                    av = v > 0.0f ? v : 0.0f;
                    d = v > 0.0f ? 1.0f : 0.0f;
                } else if (kind == BG_ACT_TANH) {
                    av = tanhf(v);
                    d = 1.0f - av * av;
                }
            };
            if constexpr (skip(8192)) lap(4);
            const bool swapfold = (NRT == 6 || NRT == 1) && span == 64 && !skip(2048);      // workgroup-uniform
            if (swapfold) {
                // Across the four rows of 16 lanes with v_permlane16_swap / v_permlane32_swap (VALU; ds_bpermute would make
                // the LDS pipe the bottleneck).  One swap + add folds two values, and the pairing is chosen so that row rho of
                // the wave ends up with the totals of outputs 2 rho, 2 rho + 1 of the thread's eight, for ALL 1 + n rows:
                // the value and its tangent rows meet in one lane, which applies bias, activation and derivative scaling
                // in registers -- no second stage, one barrier per layer.
                float res[NRT][2];
#pragma unroll
                for (int r = 0; r < NRT; ++r) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float s01 = swap16_add(acc[r][e], acc[r][2 + e]);
                        const float s23 = swap16_add(acc[r][4 + e], acc[r][6 + e]);
                        res[r][e] = swap32_add(s01, s23);
                    }
                }
                if ((lane & 15) >= 16 - P && ogr < ogn) {
                    float outv[NRT][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int o = o0 + e;
                        const bool real = o < out;
                        float av, d;
                        activate(res[0][e] + (real ? (e ? bias_1 : bias_0) : 0.0f), av, d);
                        outv[0][e] = real ? av : 0.0f;                  // the padding outputs feed the next layer's padded inputs
#pragma unroll
                        for (int r = 1; r < NRT; ++r) outv[r][e] = real ? ((kind != BG_ACT_NONE) ? res[r][e] * d : res[r][e]) : 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < NRT; ++r)
                        *reinterpret_cast<float2*>(&s_act[cur ^ 1][r][o0]) = make_float2(outv[r][0], outv[r][1]);
                }
            } else {
                if (span > 16) fold([](float v) { return __shfl_xor(v, 16); });
                if (span > 32) fold([](float v) { return __shfl_xor(v, 32); });
                const int top = span < 16 ? span : 16;                // the totals sit in the last P lanes below `top`
                if (lane >= top - P && lane < top && ogr < ogn) {     // pre-activations of 8 outputs, all rows
#pragma unroll
                    for (int r = 0; r < NRT; ++r) {
                        *reinterpret_cast<float4*>(&s_act[cur ^ 1][r][4 * og]) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
                        *reinterpret_cast<float4*>(&s_act[cur ^ 1][r][half_og]) = make_float4(acc[r][4], acc[r][5], acc[r][6], acc[r][7]);
                    }
                }
            }
            if constexpr (skip(8192)) lap(5);
            __syncthreads();
            if constexpr (skip(8192)) lap(6);
            cur ^= 1;
            if (!swapfold) {                                          // second stage, one output per thread
                if (tid < ldw && !skip(4096)) {
                    const bool real = tid < out;
                    float av, d;
                    activate(s_act[cur][0][tid] + (real ? bias_v : 0.0f), av, d);
                    if (!real) { av = 0.0f; d = 0.0f; }
                    s_act[cur][0][tid] = av;
                    if (kind != BG_ACT_NONE || !real) {
#pragma unroll
                        for (int r = 1; r < NRT; ++r) s_act[cur][r][tid] *= d;
                    }
                }
                __syncthreads();
            }
            if constexpr (skip(8192)) lap(7);
            else if constexpr (kTiming) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i == l) lap(3 + i);
            }
        }
        // the sweep's right-hand matrix: row j = mode j = [d c_j / d q_p (n columns) | c_j | zeros]; modes 0 .. n-1 are the
        // primary ones (c = q_p, derivative = identity), modes n .. n+nbar-1 the closure outputs, the rest (up to m8) zero
        // (all reads unconditional, all choices selects: the branchy form of this table took 4.7 k clocks)
        if (tid < m8 + 4 * kSweepDepth) {
            const int j = tid - n;
            const bool prim = tid < n, sec = j >= 0 && j < nbar;
            const int jj = sec ? j : 0;
            float rv[NRT];
#pragma unroll
            for (int r = 0; r < NRT; ++r) rv[r] = s_act[cur][r][jj];
            const double qv = s_q[tid & (RW - 1)];
            const double cv = prim ? qv : (sec ? (double)rv[0] : 0.0);
#pragma unroll
            for (int c = 0; c < kBW; ++c) {
                double dv = 0.0;
                if constexpr (true) {
                    if (c < ANN_MAX_N && 1 + c < NRT) {                  // compile-time
                        const double d = prim ? (tid == c ? 1.0 : 0.0) : (sec ? (double)rv[(1 + c < NRT) ? 1 + c : 0] : 0.0);
                        dv = c < n ? d : 0.0;
                    }
                }
                dv = (c == n) ? cv : dv;
                s_dN[tid][c] = dv;
            }
        }
        __syncthreads();
        lap(11);
    };
    // value_only: an evaluation that feeds the decode alone leaves the n tangent rows -- 5/6 of the layer arithmetic at n = 5 --
    // out; row 0 is computed by the same operations in the same order either way (same slices, same fold).
    auto mlp = [&](bool value_only) __attribute__((always_inline)) {
        if (value_only) mlp_impl(std::integral_constant<int, 1>{});
        else if (nr <= 6) mlp_impl(std::integral_constant<int, 6>{});
        else mlp_impl(std::integral_constant<int, ANN_MAX_ROWS>{});
    };

    // ---- one sweep over [U_p | U_s]: T = U . B with B = s_dN gives the tangent W = U_p + U_s dN (columns 0 .. n-1, into
    // s_W) and the decode u = U_p q_p + U_s N(q_p) (column n, into s_u) together, on v_mfma_f64_4x4x4_4b: block = 4 mesh
    // rows, A = U^T[4 modes][those rows] straight from L2 (one double per lane), B = 4 modes x 4 columns from LDS -- one
    // 512-byte read per (4 modes, 4 columns) for the whole wave instead of a broadcast 16-byte read per lane, mode and
    // column pair, which made the LDS pipe the limit of the VALU form of this sweep (2.1 k clocks per 8 modes).
    auto closure_impl = [&](auto ncb_c, auto full_c, bool decode) __attribute__((always_inline)) {
        constexpr int NCB = decltype(ncb_c)::value;                            // column blocks compiled in: n derivatives + 1 coefficient
        constexpr bool FULL = decltype(full_c)::value;                         // N == NPAD: no ragged last lanes
        const int ak = lane >> 4, ablk = (lane >> 2) & 3, aij = lane & 3;      // A / B operand lane: 16 k + 4 blk + (i | j)
        const int wrow = 16 * S * w;                                           // first mesh row of this wave
        // MFMA number tl of a trip works on the rows wrow + S (4 blk + i) + tl: a lane's S rows are consecutive, so its share
        // of 4 modes x 16 S rows is S / 2 16-byte loads, and 16 neighbouring lanes read 128 S contiguous bytes of a mode
        const int lrow = wrow + S * (4 * ablk + aij);
        auto fetch = [&](int kc, double (&dst)[S]) {
            const double* src = a.UT + (size_t)(4 * kc + ak) * N;
            if constexpr (FULL) {
#pragma unroll
                for (int tl = 0; tl < S; tl += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(src + lrow + tl);
                    dst[tl] = v.x; dst[tl + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int tl = 0; tl < S; ++tl) dst[tl] = src[lrow + tl < N ? lrow + tl : N - 1];   // dropped below
            }
        };
        double acc[S][NCB];
#pragma unroll
        for (int tl = 0; tl < S; ++tl)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[tl][cb] = 0.0;
        auto mma = [&](int kc, const double (&av)[S]) {
            double bv[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bv[cb] = s_dN[4 * kc + ak][4 * cb + aij];
#pragma unroll
            for (int tl = 0; tl < S; ++tl)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[tl][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[tl], bv[cb], acc[tl][cb], 0, 0, 0);
        };
        // groups of 4 modes, a ring of DEPTH register buffers: DEPTH - 1 groups in flight while one is multiplied.  Three
        // is enough: the sweep moves 393 KB per evaluation at 24 B/clk per workgroup, with two workgroups on the CU close to
        // what the 64 B/clk vector-memory return path gives; six buffers measured 7 % slower.  The trip count is rounded up to a multiple of DEPTH (the extra groups meet zero rows of B; their fetches
        // are clamped to the last real group): no guards anywhere in the loop.
        constexpr int DEPTH = kSweepDepth;
        const int kcn = skip(256) ? 0 : m8 / 4;                                // (256: timing builds only)
        const int kcl = kcn - 1;
        double ab[DEPTH][S];
        if (kcn > 0) {
#pragma unroll
            for (int d = 0; d < DEPTH - 1; ++d) fetch(d < kcl ? d : kcl, ab[d]);
        }
        for (int kc = 0; kc < kcn; kc += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int nx = kc + d + DEPTH - 1;
                fetch(nx < kcl ? nx : kcl, ab[(d + DEPTH - 1) % DEPTH]);
                mma(kc + d, ab[d]);
            }
        }
        // result lane: 16 i + 4 blk + j holds T[row S (4 blk + i) + tl][column 4 cb + j]
        const int di = lane >> 4, dblk = (lane >> 2) & 3, dj = lane & 3;
#pragma unroll
        for (int tl = 0; tl < S; ++tl) {
            const int row = wrow + S * (4 * dblk + di) + tl;
            const bool in = row < N;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int col = 4 * cb + dj;
                const double v = in ? acc[tl][cb] : 0.0;
                if (col < n) s_W[row + 1][col] = v;
                if (col == n && decode) s_u[row + 2] = v;
            }
        }
        __syncthreads();
    };
    auto closure = [&](bool decode) __attribute__((always_inline)) {
        using T = std::true_type; using F = std::false_type;
        if (n <= 7) { if (N == NPAD) closure_impl(std::integral_constant<int, 2>{}, T{}, decode); else closure_impl(std::integral_constant<int, 2>{}, F{}, decode); }
        else { if (N == NPAD) closure_impl(std::integral_constant<int, 3>{}, T{}, decode); else closure_impl(std::integral_constant<int, 3>{}, F{}, decode); }
    };

#if BG_ANN_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int slot = blockIdx.x; slot < a.B; slot += gridDim.x) {
        const int smp = a.order ? a.order[slot] : slot;
        const double mu1 = a.mu1[smp], mu2 = a.mu2[smp];
        if constexpr (kTiming) tick = (long long)__builtin_amdgcn_s_memtime();
        double* hist = a.hist + (size_t)smp * (size_t)(a.nsteps + 1) * (size_t)N;
        __syncthreads();
        // ---- per-sample constants (compute_forcing_vector :427-461, f_gp of :556-558) and the initial state ----------
        double fdt[S / 4];                          // dt F of this thread's rows i = tid (+ 256)
#pragma unroll
        for (int ii = 0; ii < S / 4; ++ii) {
            const int i = tid + 256 * ii;
            double frPrev = 0.0, fl = 0.0, hf = 0.0, u = 0.0;
            if (i < N) {
                if (i > 0) {
                    const double xl = a.x[i - 1], xr = a.x[i];
                    const double he = a.nonuniform ? xr - xl : h;
                    const double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
                    const double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
                    frPrev = (f1 * GP_B + f2 * GP_A) * (0.5 * he);
                }
                if (i < N - 1) {
                    const double xl = a.x[i], xr = a.x[i + 1];
                    const double he = a.nonuniform ? xr - xl : h;
                    const double f1 = 0.02 * exp(mu2 * (GP_A * xl + GP_B * xr));
                    const double f2 = 0.02 * exp(mu2 * (GP_B * xl + GP_A * xr));
                    fl = (f1 * GP_A + f2 * GP_B) * (0.5 * he);
                    hf = he * (f1 + f2);
                }
                u = a.u0[(size_t)smp * N + i];
                hist[i] = u;
            }
            fdt[ii] = a.dt * (frPrev + fl);
            s_h[i] = hf;
            s_u[i + 2] = u;
        }
        __syncthreads();

        int flags = 0, info_out = 0;
        bool have_tangent = false;                 // s_W holds U_p + U_s dN at the float32 input s_qlast
        for (int step = 0; step < a.nsteps && info_out == 0; ++step) {
            // ---- g = M u^n + dt F (:1214) and q_p = U_p^T u^n (:1197) --------------------------------------------------
            {
                double part[ANN_MAX_N];
#pragma unroll
                for (int c = 0; c < ANN_MAX_N; ++c) part[c] = 0.0;
#pragma unroll
                for (int ii = 0; ii < S / 4; ++ii) {
                    const int i = tid + 256 * ii;
                    double g = 0.0;
                    if (i < N) {
                        const double um = s_u[i + 1], uc = s_u[i + 2], ur = s_u[i + 3];
                        if (a.nonuniform) {
                            double v = 0.0;
                            if (i > 0) v = (a.x[i] - a.x[i - 1]) / 6.0 * __builtin_fma(2.0, uc, um);
                            if (i < N - 1) v = __builtin_fma((a.x[i + 1] - a.x[i]) / 6.0, __builtin_fma(2.0, uc, ur), v);
                            g = v + fdt[ii];
                        } else {
                            double acc;
                            if (i == 0) acc = __builtin_fma(2.0, uc, ur);
                            else if (i == N - 1) acc = __builtin_fma(2.0, uc, um);
                            else acc = __builtin_fma(4.0, uc, um) + ur;
                            g = __builtin_fma(h / 6.0, acc, fdt[ii]);
                        }
#pragma unroll
                        for (int c = 0; c < ANN_MAX_N; ++c)
                            if (c < n) part[c] = __builtin_fma(a.UT[(size_t)c * N + i], uc, part[c]);
                    }
                    s_g[i] = g;
                }
#pragma unroll
                for (int c = 0; c < ANN_MAX_N; ++c) {
                    if (c < n) {
                        const double sm = wave_sum(part[c]);
                        if (lane == 0) s_part[w][c] = sm;
                    }
                }
                __syncthreads();
                if (tid < RW) s_q[tid] = (tid < n) ? (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]) : 0.0;
                __syncthreads();
            }
            // The tangent of the first pass is dN at (float) U_p^T u^n (:1197, :1219).  u^n is the decode of the previous
            // step's last q_p, so U_p^T u^n is that q_p up to 1e-15 and its float32 image is, as a rule, the very input of the
            // previous step's last evaluation -- whose tangent is still in s_W.  When the eight floats are bitwise equal the
            // evaluation would reproduce it bit for bit and is skipped; otherwise (first step, a rounding boundary) it runs.
            bool skip_eval = have_tangent && !kTiming && !a.no_reuse;
#pragma unroll
            for (int c = 0; c < RW; ++c) skip_eval = skip_eval && ((c < n ? (float)s_q[c] : 0.0f) == s_qlast[c]);
            lap(13);
            int k = 0;
            bool more = true, decode = false;
            while (true) {
                rederive();
                // ---- closure at the current q_p (one call site: the code is inlined once).  First pass of a time step:
                // dN at the first guess (:1219), tangent only, U0 stays u^n.  Later passes: q_s = N(q_p) for the decode
                // (:1241-1242) and dN for the next projection (:1219-1224).
                lap(2);
                if (!skip_eval) {
                    // the last evaluation of the last step feeds the decode alone: value only
                    const bool value_only = !more && step == a.nsteps - 1 && !kTiming;
#if BG_ANN_PRIO & 2
                    __builtin_amdgcn_s_setprio(0);
#endif
                    mlp(value_only);
#if BG_ANN_PRIO & 1
                    __builtin_amdgcn_s_setprio(0);
#endif
                    closure(decode);
#if BG_ANN_PRIO
                    __builtin_amdgcn_s_setprio(3);
#endif
                    have_tangent = !value_only;
                }
                skip_eval = false;
                lap(12);
                if (!more) break;
                decode = true;
                // ---- assembly: A(u_k), R(u_k) per row into LDS ----------------------------------------------------
                for (int i = tid; i < (skip(16) ? 0 : NPAD); i += 256) {
                    double lo, di, up, R;
                    const bool in = i < N;
                    const MeshConst mc = make_mesh_const(h, a.dt, a.E, a.supg);
                    rom_assemble_row(i, N, s_u[i + 1], s_u[i + 2], (i + 1 < N) ? s_u[i + 3] : 0.0, in ? s_g[i] : 0.0,
                                     (in && i > 0) ? s_h[i - 1] : 0.0, (in && i < N - 1) ? s_h[i] : 0.0, mu1, mc,
                                     a.nonuniform, a.x, a.dt, a.E, lo, di, up, R);
                    s_coef[i][0] = lo; s_coef[i][1] = di; s_coef[i][2] = up; s_coef[i][3] = R;
                }
                // ---- tangent fragments of this lane from LDS (the tangent changes every iteration) ----------------
                double frag[NB][S];
#pragma unroll
                for (int c = 0; c < NB; ++c) {
#pragma unroll
                    for (int s = 0; s < S; ++s) frag[c][s] = s_W[rowbase + s + 1][4 * c + t];
                    s_halo[0][c][tid] = s_W[rowbase][4 * c + t];
                    s_halo[1][c][tid] = s_W[rowbase + S + 1][4 * c + t];
                }
                __syncthreads();
                lap(0);
#if BG_ANN_PRIO & 4
                __builtin_amdgcn_s_setprio(0);
#endif
                if constexpr (!skip(1))
                    mfma_pass<S, NB, GAL, 0, NB, true, RW>(frag, HaloTable<NB>{s_halo, tid}, s_coef, s_u, rowbase, t, w, lane, s_red, s_wtu);
#if BG_ANN_PRIO & 4
                __builtin_amdgcn_s_setprio(3);
#endif
                __syncthreads();
                lap(1);
                // ---- reduced solve -------------------------------------------------------------------------------
                // np.linalg.solve's partial-pivoting elimination by wave 0 (pivoted_gj8).  The guarded pivot-free
                // elimination of bg_rom_run does not apply here: the columns of U_p + U_s dN are far from orthonormal, the
                // multipliers exceed 1 in nearly every system, and LAPACK does leave the diagonal.
                if (tid == 0) s_info = 0;
                __syncthreads();
                if (w == 0 && !skip(2)) pivoted_gj8<GAL>(s_red, s_x, &s_info, lane, n);
                __syncthreads();
                const double xout = (lane < RW) ? s_x[lane] : 0.0;
                if (s_info != 0 && info_out == 0) info_out = s_info;
                // ---- q_p += dq, err = |dq| / (|q_p| + 1e-14)  (:1238-1244) -----------------------------------------
                const double dq = (lane < n) ? xout : 0.0;
                const double qn = (lane < n) ? s_q[lane] + dq : 0.0;
                double nd, nq;
                wave_sum2(dq * dq, qn * qn, nd, nq);
                nd = sqrt(nd); nq = sqrt(nq);
                const double err = nd / (nq + 1e-14);
                ++k;
                more = (err > a.tol) && (k < a.max_it) && info_out == 0;
                if (kTiming) more = k < 5;
                if (!(err - err == 0.0)) flags |= BG_FLAG_NONFINITE;
                if (k >= a.max_it) flags |= BG_FLAG_HIT_CAP;
                __syncthreads();                                   // every wave has read s_q
                if (w == 0 && lane < RW) s_q[lane] = qn;
                __syncthreads();
            }
            double* hrow = hist + (size_t)(step + 1) * N;
            for (int i = tid; i < N; i += 256) hrow[i] = s_u[i + 2];
            if (tid == 0) a.iters[(size_t)smp * a.nsteps + step] = k;
        }
        if (tid == 0) {
            a.flags[smp] = flags;
            a.info[smp] = info_out;
        }
        if (kTiming && tid == 0 && a.nsteps >= 16) {     // diagnostic builds only: kilo-clocks per phase in place of the counts
#pragma unroll
            for (int i = 0; i < 16; ++i) a.iters[(size_t)smp * a.nsteps + i] = (int)(cyc[i] >> 10);
#pragma unroll
            for (int i = 0; i < 16; ++i) cyc[i] = 0;
        }
    }
}

template <int S>
void launch_ann(int projection, int grid, hipStream_t st, const AnnRunArgs& a)
{
    if (projection == BG_PROJ_GALERKIN)
        hipLaunchKernelGGL((rom_ann_fused_kernel<S, BG_PROJ_GALERKIN>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((rom_ann_fused_kernel<S, BG_PROJ_LSPG>), dim3(grid), dim3(256), 0, st, a);
}

}  // namespace

extern "C" {

int bg_ann_rom_limits(int* max_n, int* max_nbar, int* max_width, int* max_layers)
{
    if (max_n) *max_n = ANN_MAX_N;
    if (max_nbar) *max_nbar = 128;
    if (max_width) *max_width = ANN_MAX_WIDTH;
    if (max_layers) *max_layers = ANN_MAX_LAYERS;
    return BG_OK;
}

int bg_ann_rom_run(int N, int B, int n, int nbar, int nsteps, int projection, const double* x, const double* UT,
                   const double* u0, const double* mu1, const double* mu2, int n_layers,
                   const int* widths, const float* const* wt, const float* const* bias, const int* acts,
                   const float* alphas, double dt, double E, double tol, int max_it, int options, double* hist,
                   int32_t* iters, int32_t* flags, int32_t* info, const int32_t* order, void* stream)
{
    if (N < 2 || B < 0 || n < 1 || nbar < 1 || nsteps < 0 || max_it < 1 || !(dt > 0.0) || n_layers < 1) return BG_ERR_BAD_ARG;
    if (projection != BG_PROJ_GALERKIN && projection != BG_PROJ_LSPG) return BG_ERR_PROJECTION;
    if (N > 512) return BG_ERR_UNSUPPORTED_N;
    if (n > ANN_MAX_N || nbar > 128 || n_layers > ANN_MAX_LAYERS) return BG_ERR_UNSUPPORTED_R;
    if (!widths || !wt || !bias || !acts || !alphas) return BG_ERR_BAD_ARG;
    if (widths[0] != n || widths[n_layers] != nbar) return BG_ERR_BAD_ARG;
    AnnRunArgs a;
    for (int l = 0; l < n_layers; ++l) {
        if (widths[l + 1] < 1 || widths[l + 1] > ANN_MAX_WIDTH) return BG_ERR_UNSUPPORTED_R;
        if (!wt[l] || ((uintptr_t)wt[l] & 15)) return BG_ERR_BAD_ARG;
        if (acts[l] != BG_ACT_NONE && acts[l] != BG_ACT_ELU && acts[l] != BG_ACT_RELU && acts[l] != BG_ACT_TANH) return BG_ERR_BAD_ARG;
        a.wt[l] = wt[l]; a.bias[l] = bias[l]; a.act[l] = acts[l]; a.alpha[l] = alphas[l];
    }
    for (int l = 0; l <= n_layers; ++l) a.width[l] = widths[l];
    if (B == 0) return BG_OK;
    if (!x || !UT || !u0 || !mu1 || !mu2 || !hist || !flags || !info || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    a.x = x; a.UT = UT; a.u0 = u0; a.mu1 = mu1; a.mu2 = mu2; a.hist = hist; a.iters = iters; a.flags = flags;
    a.info = info; a.order = order; a.nl = n_layers; a.dt = dt; a.E = E; a.tol = tol; a.N = N; a.B = B; a.n = n; a.nbar = nbar;
    a.nsteps = nsteps; a.max_it = max_it; a.supg = options & BG_OPT_SUPG; a.nonuniform = (options & BG_OPT_NONUNIFORM) ? 1 : 0;
    a.no_reuse = (options & BG_OPT_NO_TANGENT_REUSE) ? 1 : 0;
    const int slots = 2 * device_cu_count();         // two workgroups per CU: one's memory latency hides behind the other
    const int grid = B < slots ? B : slots;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 256) launch_ann<4>(projection, grid, st, a); else launch_ann<8>(projection, grid, st, a);
    return check_launch();
}

}  // extern "C"
