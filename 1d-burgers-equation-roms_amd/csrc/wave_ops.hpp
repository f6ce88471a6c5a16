// wave_ops.hpp -- wave64 cross-lane and arithmetic helpers for gfx950 (CDNA4).
//
// One wavefront owns one Burgers sample; everything below is wave-private (no LDS
// storage, no barriers).  Cross-lane traffic uses DPP moves where the pattern allows
// (1 VALU op per dword) and ds_bpermute (LDS crossbar, no LDS memory) otherwise.
#pragma once
#include <hip/hip_runtime.h>

namespace bg {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---- DPP moves on doubles ---------------------------------------------------------
// dpp_ctrl encodings (GFX9): wave_shl:1 = 0x130, wave_shr:1 = 0x138,
// row_shr:n = 0x110+n, row_bcast15 = 0x142, row_bcast31 = 0x143.
template <int CTRL, int ROW_MASK = 0xF, bool BOUND_ZERO = true>
__device__ __forceinline__ double dpp_mov(double v, double old = 0.0)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int olo = __double2loint(old), ohi = __double2hiint(old);
    lo = __builtin_amdgcn_update_dpp(olo, lo, CTRL, ROW_MASK, 0xF, BOUND_ZERO);
    hi = __builtin_amdgcn_update_dpp(ohi, hi, CTRL, ROW_MASK, 0xF, BOUND_ZERO);
    return __hiloint2double(hi, lo);
}

// value held by lane-1 (lane 0 receives 0)
__device__ __forceinline__ double from_lane_below(double v) { return dpp_mov<0x138>(v); }
// value held by lane+1 (lane 63 receives 0)
__device__ __forceinline__ double from_lane_above(double v) { return dpp_mov<0x130>(v); }

// value held by lane (lane + delta) mod 64, any delta; ds_bpermute, 2 ops per double
__device__ __forceinline__ double from_lane_rot(double v, int src_lane_times4)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src_lane_times4, lo);
    hi = __builtin_amdgcn_ds_bpermute(src_lane_times4, hi);
    return __hiloint2double(hi, lo);
}

// ---- wave-wide sum, result broadcast to every lane --------------------------------
// row_shr 1,2,4,8 inside each row of 16, then row_bcast15 / row_bcast31 across rows;
// lane 63 ends up with the total, which is then read back through an SGPR pair.
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov<0x111>(v);            // row_shr:1
    v += dpp_mov<0x112>(v);            // row_shr:2
    v += dpp_mov<0x114>(v);            // row_shr:4
    v += dpp_mov<0x118>(v);            // row_shr:8
    v += dpp_mov<0x142, 0xA>(v);       // row_bcast15 -> rows 1,3
    v += dpp_mov<0x143, 0xC>(v);       // row_bcast31 -> rows 2,3
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// ---- reciprocal: v_rcp_f64 seed (~2^-23 relative) + two Newton steps ---------------
__device__ __forceinline__ double rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// one Newton step only: 2.1e-15 max relative error measured on gfx950 (tools/microbench.hip);
// used where the consumer is itself an iteration that contracts such perturbations
__device__ __forceinline__ double rcp1(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}

// DPP move applied REPS times (wave_shr:1 twice = shift by two lanes, zero filled)
template <int CTRL, int REPS>
__device__ __forceinline__ double dpp_shift(double v)
{
#pragma unroll
    for (int i = 0; i < REPS; ++i) v = dpp_mov<CTRL>(v);
    return v;
}

// two wave-wide sums for little more than the price of one: v_permlane32_swap puts the
// partial sums of `a` in lanes 0..31 and those of `b` in lanes 32..63, then one row scan.
__device__ __forceinline__ void wave_sum2(double a, double b, double& sa, double& sb)
{
    auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    double v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    v += dpp_mov<0x111>(v);            // row_shr:1
    v += dpp_mov<0x112>(v);            // row_shr:2
    v += dpp_mov<0x114>(v);            // row_shr:4
    v += dpp_mov<0x118>(v);            // row_shr:8
    v += dpp_mov<0x142, 0xA>(v);       // row_bcast15 -> rows 1,3
    sa = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 31),
                          __builtin_amdgcn_readlane(__double2loint(v), 31));
    sb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                          __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ double sel(bool c, double a, double b) { return c ? a : b; }

}  // namespace bg
