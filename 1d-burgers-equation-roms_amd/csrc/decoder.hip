// decoder.hip -- the contraction of the non-intrusive POD-ANN decoder, bf16 tier (BASELINE config 5, decoder-only form).
//
// reference: Non-Instrusive/predict_pod_ann.py:73-80, `Uhat = U_modes @ Qhat.T` for every (mu1, mu2, t) column.  The MLP that
// produces Qhat stays in PyTorch-ROCm (bf16), as the config prescribes; this kernel is the dense product that follows it,
// with the result written ONCE, as the float64 snapshot layout the reference returns:
//     out[b][i][t] = sum_k Um[i][k] * Q[b * Nt + t][k]          bf16 operands, float32 accumulate (v_mfma_f32_32x32x16_bf16)
// The product is write-bound (8 N bytes per column against 2 n N flops at n = 160): a library bf16 GEMM followed by a cast
// writes the result twice (bf16, then float64) and reads it once more, 10 N + 2 N bytes per column instead of 8 N.
// Workgroup = 128 columns x all N rows: the columns' coefficients sit in LDS (40 KB at n = 160), a wave takes every fourth
// 32-row tile, its A fragments (U_modes rows, L2-resident) in registers, and writes the tile through a per-wave LDS staging
// block as runs of 128 consecutive doubles of a sample's time axis (1 KB per row, 16 bytes per lane).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"

namespace {

using namespace bg;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DEC_COLS = 128;        // columns per workgroup
constexpr int DEC_MAX_KB = 16;       // n <= 256

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
    const long long c0 = (long long)blockIdx.x * DEC_COLS;
    // the workgroup's coefficient rows (zero beyond the last column)
    for (int e = tid; e < DEC_COLS * (n / 8); e += 256) {
        const int col = e / (n / 8), ch = e - col * (n / 8);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c0 + col < C) v = *reinterpret_cast<const uint4*>(Q + (size_t)(c0 + col) * n + 8 * ch);
        *reinterpret_cast<uint4*>(&s_q[col * LD + 8 * ch]) = v;
    }
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
    __syncthreads();
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
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double2 val = *reinterpret_cast<const double2*>(&stage[r * DEC_COLS + 2 * lane]);
                const long long ro = (long long)(i0 + 8 * v4 + r) * Nt;
                if (pair) {
                    *reinterpret_cast<double2*>(out + o0 + ro) = val;       // 8-byte aligned: see above
                } else {
                    if (o0 >= 0) out[o0 + ro] = val.x;
                    if (o1 >= 0) out[o1 + ro] = val.y;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int KB>
int launch_decode(const uint16_t* Um, const uint16_t* Q, double* out, int N, int Nt, long long C, hipStream_t st)
{
    const size_t lds = (size_t)DEC_COLS * (16 * KB + 8) * sizeof(uint16_t) + 4 * 8 * DEC_COLS * sizeof(double);
    const long long grid = (C + DEC_COLS - 1) / DEC_COLS;
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
