// decoder.hip -- the contraction of the non-intrusive POD-ANN decoder, bf16 tier (BASELINE config 5, decoder-only form).
//
// reference: Non-Instrusive/predict_pod_ann.py:60-80: standardised (mu1, mu2, t) -> MLP -> Qhat, then `Uhat = U_modes @ Qhat.T`
// for every (mu1, mu2, t) column.  Two entry points:
//   bg_decode_modes_bf16   the dense product alone (Qhat from any model, e.g. the PyTorch-ROCm bf16 module), result written
//                          ONCE as the float64 snapshot layout the reference returns:
//       out[b][i][t] = sum_k Um[i][k] * Q[b * Nt + t][k]      bf16 operands, float32 accumulate (v_mfma_f32_32x32x16_bf16)
//   bg_decode_mlp_bf16     the same product with a plain MLP evaluated per workgroup in front of it (decode_mlp_kernel below):
//                          no activation and no coefficient crosses HBM, coefficients bitwise those of the PyTorch bf16 module
// The product is write-bound (8 N bytes per column against 2 n N flops at n = 160): a library bf16 GEMM followed by a cast
// writes the result twice (bf16, then float64) and reads it once more, 10 N + 2 N bytes per column instead of 8 N.
// Workgroup = 128 columns x all N rows: the columns' coefficients sit in LDS (40 KB at n = 160), a wave takes every fourth
// 32-row tile, its A fragments (U_modes rows, L2-resident) in registers, and writes the tile through a per-wave LDS staging
// block as runs of 128 consecutive doubles of a sample's time axis (1 KB per row, 16 bytes per lane); the chunks are dealt to the
// XCDs so that one L2 collects the pieces of a sample's rows (xcd_chunk).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"

namespace {

using namespace bg;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DEC_COLS = 128;        // columns per workgroup
constexpr int DEC_MAX_KB = 16;       // n <= 256

// XCD-aware order of the 128-column chunks.  Workgroups go round-robin to the 8 XCDs (each with its own L2), so with
// chunk = blockIdx the four chunks that make up the 4-KB rows of ONE sample are written through four different L2s at
// unrelated times; chunk = (blockIdx % 8) * (chunks per XCD) + blockIdx / 8 puts neighbouring chunks on the same XCD in
// neighbouring dispatch slots: one L2 sees a row's pieces together.  As pure stores (tools/store_pattern_bench.hip,
// 2.1 GB): 2.97 -> 3.64 TB/s for this tile shape; 64 KB contiguous per workgroup reaches 5.3 -- 5.7, out of reach here
// (the coefficients of all 501 columns of a sample do not fit LDS: 164 KB).
__device__ __forceinline__ long long xcd_chunk(long long C)
{
    const long long per = (gridDim.x + 7) / 8;
    (void)C;
    return (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
}

// The contraction proper, shared by the two kernels: s_q holds the workgroup's 128 coefficient rows ([DEC_COLS][LD] bf16), the
// caller has synchronised.
template <int KB>
__device__ __forceinline__ void decode_tiles(const uint16_t* __restrict__ Um, const uint16_t* s_q, const int LD, double* stage,
                                             double* __restrict__ out, int N, int Nt, long long C, long long c0, int lane, int w)
{
    constexpr int n = 16 * KB;
    // operand lanes of v_mfma_f32_32x32x16_bf16: lane l holds row (A) / column (B) l % 32, k = 8 (l / 32) .. + 7;
    // result: acc[v] = D[8 (v / 4) + 4 (l / 32) + v % 4][l % 32]
    const int lr = lane & 31, lh = lane >> 5;
    // write-out role: lane l owns columns 2 l, 2 l + 1 of the workgroup's 128 (a column is one (sample, time level) pair)
    long long o0, o1;                       // element offsets of the two columns at row 0 (-1: beyond the last column)
    {
        const long long ca = c0 + 2 * lane, cb = ca + 1;
        const long long ba = (ca < C ? ca : 0) / Nt, bb = (cb < C ? cb : 0) / Nt;
        o0 = ca < C ? ba * (long long)N * Nt + (ca - ba * Nt) : -1;
        o1 = cb < C ? bb * (long long)N * Nt + (cb - bb * Nt) : -1;
    }
    const bool pair = o0 >= 0 && o1 == o0 + 1;      // both columns in the same sample: one 16-byte store per row
    const bool all_pair = __ballot(pair) == ~0ull;  // no sample boundary and no ragged end inside this wave's 128 columns
    for (int it = w; it < N / 32; it += 4) {
        const int i0 = 32 * it;
        bf16x8 a[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            a[kb] = *reinterpret_cast<const bf16x8*>(Um + (size_t)(i0 + lr) * n + 16 * kb + 8 * lh);
        f32x16 acc[DEC_COLS / 32];
#pragma unroll
        for (int ts = 0; ts < DEC_COLS / 32; ++ts) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ts][v] = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bf16x8 bq = *reinterpret_cast<const bf16x8*>(&s_q[(32 * ts + lr) * LD + 16 * kb + 8 * lh]);
                acc[ts] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb], bq, acc[ts], 0, 0, 0);
            }
        }
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {             // rows i0 + 8 v4 .. + 7 of all 128 columns
#pragma unroll
            for (int ts = 0; ts < DEC_COLS / 32; ++ts) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    stage[(4 * lh + e) * DEC_COLS + 32 * ts + lr] = (double)acc[ts][4 * v4 + e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (all_pair) {                              // (wave-uniform) the common case: eight reads in flight, eight stores
                double2 val[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) val[r] = *reinterpret_cast<const double2*>(&stage[r * DEC_COLS + 2 * lane]);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    *reinterpret_cast<double2*>(out + o0 + (long long)(i0 + 8 * v4 + r) * Nt) = val[r];   // 8-byte aligned: see above
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const double2 val = *reinterpret_cast<const double2*>(&stage[r * DEC_COLS + 2 * lane]);
                    const long long ro = (long long)(i0 + 8 * v4 + r) * Nt;
                    if (pair) {
                        *reinterpret_cast<double2*>(out + o0 + ro) = val;
                    } else {
                        if (o0 >= 0) out[o0 + ro] = val.x;
                        if (o1 >= 0) out[o1 + ro] = val.y;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int KB>                    // KB = n / 16
__global__ __launch_bounds__(256, 2) void decode_modes_kernel(const uint16_t* __restrict__ Um, const uint16_t* __restrict__ Q,
                                                           double* __restrict__ out, int N, int Nt, long long C)
{
    constexpr int n = 16 * KB, LD = n + 8;                       // LDS row stride in bf16: 16-byte aligned, conflict-light
    extern __shared__ __attribute__((aligned(16))) uint16_t s_q[];   // [DEC_COLS][LD], then the per-wave staging rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Staging: 8 result rows x 128 columns of float64 per wave.  The accumulators of a 32 x 32 MFMA tile hold, per lane, ONE
    // column and sixteen rows: stored as they stand, an instruction writes two 256-byte pieces with 8 bytes per lane (round 2:
    // 3.0 TB/s).  Through LDS a wave turns eight rows of all four sub-tiles into eight runs of 1024 contiguous bytes along
    // the time axis, 16 bytes per lane (the row pitch Nt * 8 = 4008 bytes leaves the runs 8-byte aligned only; the hardware
    // splits the few lanes that straddle a line).
    double* stage = reinterpret_cast<double*>(s_q + DEC_COLS * LD) + w * (8 * DEC_COLS);
    const long long c0 = xcd_chunk(C) * DEC_COLS;
    if (c0 >= C) return;                                         // (the grid is rounded up to a multiple of 8)
    // the workgroup's coefficient rows (zero beyond the last column)
    for (int e = tid; e < DEC_COLS * (n / 8); e += 256) {
        const int col = e / (n / 8), ch = e - col * (n / 8);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c0 + col < C) v = *reinterpret_cast<const uint4*>(Q + (size_t)(c0 + col) * n + 8 * ch);
        *reinterpret_cast<uint4*>(&s_q[col * LD + 8 * ch]) = v;
    }
    __syncthreads();
    decode_tiles<KB>(Um, s_q, LD, stage, out, N, Nt, C, c0, lane, w);
}

template <int KB>
int launch_decode(const uint16_t* Um, const uint16_t* Q, double* out, int N, int Nt, long long C, hipStream_t st)
{
    const size_t lds = (size_t)DEC_COLS * (16 * KB + 8) * sizeof(uint16_t) + 4 * 8 * DEC_COLS * sizeof(double);
    const long long grid = ((C + DEC_COLS - 1) / DEC_COLS + 7) / 8 * 8;      // see xcd_chunk
    if (lds > 64 * 1024) {                   // beyond the default dynamic-LDS limit: raise it once per instantiation and device
        static std::atomic<unsigned> done{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_modes_kernel<KB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return check_launch() == BG_OK ? BG_ERR_LAUNCH : BG_ERR_LAUNCH;
            done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL((decode_modes_kernel<KB>), dim3((unsigned)grid), dim3(256), lds, st, Um, Q, out, N, Nt, C);
    return check_launch();
}


// ---- bg_decode_mlp_bf16: the decoder MLP evaluated in the same kernel ------------------------------------------------------------
// reference: Non-Instrusive/predict_pod_ann.py:60-80 -- standardised (mu1, mu2, t) -> MLP -> Qhat -> U_modes @ Qhat.T.  The PyTorch
// form of the bf16 tier runs the MLP as one GEMM + one activation launch per layer over all B Nt columns: every activation
// crosses HBM twice per layer and the coefficients once more on their way into decode_modes_kernel (40 % of a pass at config 5).
// Here a workgroup evaluates the MLP for its own 128 columns before it contracts them: wave w owns the columns 32 w .. 32 w + 31
// through ALL layers (no workgroup barrier between layers), activations as bf16 in the coefficient block of LDS, in place
// (a layer's inputs sit in registers as MFMA B fragments before its outputs overwrite them), v_mfma_f32_32x32x16_bf16 with
// A = 32 output features x 16 inputs of W straight from L2 (62 KB for the committed 3-32-64-128-160 model) -- the same
// rounding points as the PyTorch bf16 module (Linear output -> bf16, activation in float32 -> bf16).
constexpr int DEC_MAX_LAYERS = 8;
constexpr int DEC_MAX_WIDTH = 256;

struct DecMlpArgs {
    const uint16_t* W[DEC_MAX_LAYERS];       // layer l: [wout[l]][win[l]] bf16 row-major (torch Linear.weight), zero padded
    const uint16_t* bias[DEC_MAX_LAYERS];    // [wout[l]] bf16 (zero padded) or null
    int win[DEC_MAX_LAYERS];                 // padded input width: a multiple of 16 (layer 0: 16)
    int wout[DEC_MAX_LAYERS];                // padded output width: a multiple of 32
    int act[DEC_MAX_LAYERS];
    float alpha[DEC_MAX_LAYERS];
    int nl;
    const double* z1;                        // [B] standardised mu1
    const double* z2;                        // [B] standardised mu2
    const double* ztau;                      // [Nt] standardised time levels
};

__device__ __forceinline__ uint16_t f32_to_bf16(float f)          // round to nearest even, as torch's .to(bfloat16)
{
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

template <int KB, int MAXKB>          // MAXKB = the widest layer INPUT / 16 (8 or 16): sizes the register fragments
__global__ __launch_bounds__(256, 2) void decode_mlp_kernel(const uint16_t* __restrict__ Um, const DecMlpArgs m, double* __restrict__ out,
                                                           int N, int Nt, long long C, int LD)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t s_q[];   // [DEC_COLS][LD] (LD >= every layer width + 8), then the staging rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* stage = reinterpret_cast<double*>(s_q + DEC_COLS * LD) + w * (8 * DEC_COLS);
    const long long c0 = xcd_chunk(C) * DEC_COLS;
    if (c0 >= C) return;                                         // (the grid is rounded up to a multiple of 8)
    const int lr = lane & 31, lh = lane >> 5;
    uint16_t* myrow = s_q + (32 * w + lr) * LD;                       // this lane's column of the workgroup's 128
    // ---- the network input of this lane's column: bf16(z1[b]), bf16(z2[b]), bf16(ztau[t]), zeros up to 16 ----------------------
    {
        const long long c = c0 + 32 * w + lr;
        const bool valid = c < C;
        const long long b = (valid ? c : 0) / Nt;
        const int t = (int)((valid ? c : 0) - b * Nt);
        if (lh == 0) {
            const uint32_t x0 = valid ? f32_to_bf16((float)m.z1[b]) : 0u, x1 = valid ? f32_to_bf16((float)m.z2[b]) : 0u;
            const uint32_t x2 = valid ? f32_to_bf16((float)m.ztau[t]) : 0u;
            *reinterpret_cast<uint4*>(myrow) = make_uint4(x0 | (x1 << 16), x2, 0u, 0u);
            *reinterpret_cast<uint4*>(myrow + 8) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // The (layer, 32-feature tile) pairs run as ONE sequence with the weights and bias of the next pair in flight while this
    // one is multiplied and activated: they do not depend on the activations, so the prefetch crosses layer boundaries.
    // No branches inside: fragments beyond a layer's inputs re-read its last 16 (a cache hit) and meet zero B fragments --
    // 35 idle matrix instructions per 244 at the committed widths, against one exposed L2 round trip per guarded load
    // (first version, with `if (kb < nkb)` around every load + MFMA and a divergent branch around expf: 0.783 ms per 1024
    // samples; this form 0.697 ms; the contraction alone 0.637 ms -- DESIGN K6).
    auto fetch = [&](int l, int mt, bf16x8 (&a)[MAXKB], uint2 (&bz)[4]) {
        const int win = m.win[l], nkb = win >> 4;
        const uint16_t* wrow = m.W[l] + (size_t)(32 * mt + lr) * win + 8 * lh;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) a[kb] = *reinterpret_cast<const bf16x8*>(wrow + 16 * (kb < nkb ? kb : nkb - 1));
        const uint16_t* bl = m.bias[l] ? m.bias[l] : m.W[l];          // (a valid address either way; masked below)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 v = *reinterpret_cast<const uint2*>(bl + 32 * mt + 8 * g + 4 * lh);
            bz[g] = m.bias[l] ? v : make_uint2(0u, 0u);
        }
    };
    bf16x8 a_cur[MAXKB], a_nxt[MAXKB], bq[MAXKB];
    uint2 b_cur[4], b_nxt[4];
    fetch(0, 0, a_cur, b_cur);
    int l = 0, mt = 0;
    while (l < m.nl) {
        const int nkb = m.win[l] >> 4, nmt = m.wout[l] >> 5, kind = m.act[l];
        const float alpha = m.alpha[l];
        if (mt == 0) {
            // this wave's 32 columns of the layer input as B fragments (lane: column l % 32, inputs 16 kb + 8 (l / 32) .. + 7)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kb = 0; kb < MAXKB; ++kb) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(myrow + 16 * (kb < nkb ? kb : 0) + 8 * lh);
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
                bq[kb] = kb < nkb ? v : z;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                         // every lane holds its inputs: the row may be overwritten
        }
        const bool last_tile = mt + 1 == nmt;
        const int l2 = last_tile ? l + 1 : l, mt2 = last_tile ? 0 : mt + 1;
        const bool more = l2 < m.nl;
        fetch(more ? l2 : l, more ? mt2 : mt, a_nxt, b_nxt);
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[kb], bq[kb], acc, 0, 0, 0);
        // acc[v] = feature 32 mt + 8 (v / 4) + 4 (l / 32) + v % 4 of column l % 32: four consecutive features per group of 4.
        // The activation kind is wave-uniform: one switch per tile, selects inside (no divergent branches around expf).
        auto epilogue = [&](auto kind_c) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t packed[2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t bb = (uint16_t)((e < 2 ? b_cur[g].x : b_cur[g].y) >> (16 * (e & 1)));
                    // Linear: float32 accumulate + bias, rounded to bf16; activation in float32 on that value, rounded again
                    float v = bf16_to_f32(f32_to_bf16(acc[4 * g + e] + bf16_to_f32(bb)));
                    if constexpr (KIND == BG_ACT_ELU) { const float ex = alpha * (expf(fminf(v, 0.0f)) - 1.0f); v = v > 0.0f ? v : ex; }
                    else if constexpr (KIND == BG_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if constexpr (KIND == BG_ACT_TANH) v = tanhf(v);
                    const uint32_t h = f32_to_bf16(v);
                    if (e & 1) packed[e >> 1] |= h << 16; else packed[e >> 1] = h;
                }
                *reinterpret_cast<uint2*>(myrow + 32 * mt + 8 * g + 4 * lh) = make_uint2(packed[0], packed[1]);
            }
        };
        switch (kind) {
            case BG_ACT_ELU: epilogue(std::integral_constant<int, BG_ACT_ELU>{}); break;
            case BG_ACT_RELU: epilogue(std::integral_constant<int, BG_ACT_RELU>{}); break;
            case BG_ACT_TANH: epilogue(std::integral_constant<int, BG_ACT_TANH>{}); break;
            default: epilogue(std::integral_constant<int, BG_ACT_NONE>{}); break;
        }
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) a_cur[kb] = a_nxt[kb];
#pragma unroll
        for (int g = 0; g < 4; ++g) b_cur[g] = b_nxt[g];
        l = l2; mt = mt2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __syncthreads();
    // the store phase of one workgroup goes ahead of the other's MLP phase (bulk VALU + MFMA work) on the shared SIMDs: the
    // stores it feeds are what the kernel waits for (0.572 -> 0.548 ms per 1024 samples)
    __builtin_amdgcn_s_setprio(3);
    decode_tiles<KB>(Um, s_q, LD, stage, out, N, Nt, C, c0, lane, w);
}

template <int KB, int MAXKB>
int launch_decode_mlp(const uint16_t* Um, const DecMlpArgs& m, double* out, int N, int Nt, long long C, int LD, hipStream_t st)
{
    const size_t lds = (size_t)DEC_COLS * LD * sizeof(uint16_t) + 4 * 8 * DEC_COLS * sizeof(double);
    const long long grid = ((C + DEC_COLS - 1) / DEC_COLS + 7) / 8 * 8;      // see xcd_chunk
    if (lds > 64 * 1024) {                   // beyond the default dynamic-LDS limit: raise it (per instantiation; idempotent)
        static std::atomic<int> granted[32];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (granted[dev & 31].load(std::memory_order_relaxed) < (int)lds) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_mlp_kernel<KB, MAXKB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return check_launch() == BG_OK ? BG_ERR_LAUNCH : BG_ERR_LAUNCH;
            granted[dev & 31].store((int)lds, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL((decode_mlp_kernel<KB, MAXKB>), dim3((unsigned)grid), dim3(256), lds, st, Um, m, out, N, Nt, C, LD);
    return check_launch();
}

}  // namespace

extern "C" int bg_decode_modes_bf16(int N, int n, int B, int Nt, const uint16_t* Um, const uint16_t* Q, double* out, void* stream)
{
    if (N < 1 || n < 1 || B < 0 || Nt < 1) return BG_ERR_BAD_ARG;
    if (N % 32 != 0) return BG_ERR_UNSUPPORTED_N;
    if (n % 16 != 0 || n > 16 * DEC_MAX_KB) return BG_ERR_UNSUPPORTED_R;
    if (B == 0) return BG_OK;
    if (!Um || !Q || !out) return BG_ERR_BAD_ARG;
    if (((uintptr_t)Um | (uintptr_t)Q) & 15) return BG_ERR_BAD_ARG;
    const long long C = (long long)B * Nt;
    if ((C + DEC_COLS - 1) / DEC_COLS > 0x7fffffffLL) return BG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (n / 16) {
#define BG_DEC(K) case K: return launch_decode<K>(Um, Q, out, N, Nt, C, st);
        BG_DEC(1) BG_DEC(2) BG_DEC(3) BG_DEC(4) BG_DEC(5) BG_DEC(6) BG_DEC(7) BG_DEC(8)
        BG_DEC(9) BG_DEC(10) BG_DEC(11) BG_DEC(12) BG_DEC(13) BG_DEC(14) BG_DEC(15) BG_DEC(16)
#undef BG_DEC
    }
    return BG_ERR_UNSUPPORTED_R;
}

extern "C" int bg_decode_mlp_bf16(int N, int n, int B, int Nt, const uint16_t* Um, const double* z1, const double* z2, const double* ztau,
                                  int n_layers, const int* win, const int* wout, const uint16_t* const* W, const uint16_t* const* bias,
                                  const int* acts, const float* alphas, double* out, void* stream)
{
    if (N < 1 || n < 1 || B < 0 || Nt < 1 || n_layers < 1 || !win || !wout || !W || !bias || !acts || !alphas) return BG_ERR_BAD_ARG;
    if (N % 32 != 0) return BG_ERR_UNSUPPORTED_N;
    if (n % 32 != 0 || n > 16 * DEC_MAX_KB || n_layers > DEC_MAX_LAYERS) return BG_ERR_UNSUPPORTED_R;
    DecMlpArgs m;
    int maxw = n, maxin = 16;
    for (int l = 0; l < n_layers; ++l) {
        if (win[l] < 16 || win[l] % 16 != 0 || wout[l] < 32 || wout[l] % 32 != 0 || !W[l]) return BG_ERR_BAD_ARG;
        if (win[l] > DEC_MAX_WIDTH || wout[l] > DEC_MAX_WIDTH) return BG_ERR_UNSUPPORTED_R;
        if (l > 0 && win[l] != wout[l - 1]) return BG_ERR_BAD_ARG;
        if (acts[l] != BG_ACT_NONE && acts[l] != BG_ACT_ELU && acts[l] != BG_ACT_RELU && acts[l] != BG_ACT_TANH) return BG_ERR_BAD_ARG;
        if (((uintptr_t)W[l] & 15) || ((uintptr_t)bias[l] & 7)) return BG_ERR_BAD_ARG;
        m.W[l] = W[l]; m.bias[l] = bias[l]; m.win[l] = win[l]; m.wout[l] = wout[l]; m.act[l] = acts[l]; m.alpha[l] = alphas[l];
        if (win[l] > maxw) maxw = win[l];
        if (wout[l] > maxw) maxw = wout[l];
        if (win[l] > maxin) maxin = win[l];
    }
    if (win[0] != 16 || wout[n_layers - 1] != n) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!Um || !z1 || !z2 || !ztau || !out) return BG_ERR_BAD_ARG;
    if ((uintptr_t)Um & 15) return BG_ERR_BAD_ARG;
    m.nl = n_layers; m.z1 = z1; m.z2 = z2; m.ztau = ztau;
    const long long C = (long long)B * Nt;
    if ((C + DEC_COLS - 1) / DEC_COLS > 0x7fffffffLL) return BG_ERR_BAD_ARG;
    const int LD = maxw + 8;
    hipStream_t st = (hipStream_t)stream;
    switch (n / 16) {
#define BG_DEC(K) case K: return maxin <= 128 ? launch_decode_mlp<K, 8>(Um, m, out, N, Nt, C, LD, st) \
                                               : launch_decode_mlp<K, 16>(Um, m, out, N, Nt, C, LD, st);
        BG_DEC(2) BG_DEC(4) BG_DEC(6) BG_DEC(8) BG_DEC(10) BG_DEC(12) BG_DEC(14) BG_DEC(16)
#undef BG_DEC
    }
    return BG_ERR_UNSUPPORTED_R;
}
