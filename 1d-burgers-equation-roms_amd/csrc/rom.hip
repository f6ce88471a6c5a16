// rom.hip -- batched projection-ROM kernels for gfx950 and their C-ABI entry points.
//
// One Picard/Gauss-Newton iteration of the reference's ROM time-steppers
// (FEM/fem_burgers.py: pod_prom_burgers :709-785, pod_quadratic_manifold :1081-1175,
// pod_ann_prom :1177-1251) is, per sample,
//     assemble A(u), R(u)            -> fused into bg_rom_reduce (same arithmetic as the FOM)
//     Ar = W^T A W | (A W)^T (A W)   -> bg_rom_reduce, fp64 MFMA 16x16x4, W = basis / tangent
//     br = W^T R   | (A W)^T R       -> same kernel (extra B-operand column)
//     dq = solve(Ar, -br)            -> bg_lu_solve (partial pivoting, one wavefront per system)
// The decode / tangent steps that differ between the three ROM families are plain dense
// contractions over the whole batch and live on the host side (burgers_hip/rom.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "rom_device.hpp"

namespace {

using namespace bg;
using f64x4 = __attribute__((ext_vector_type(4))) double;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// ------------------------------------------------------------------------------------
// Per-sample constants: fdt[b][i] = dt*F_i(mu2_b), hfs[b][e] = h*(f(gp1)+f(gp2)) of element e
// (hfs[b][N-1] = 0).  reference: compute_forcing_vector :427-461 and the f_gp of :556-558.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void forcing_setup_kernel(const double* __restrict__ x,
                                                            const double* __restrict__ mu2, int N, int B,
                                                            double dt, int nonuniform, double* __restrict__ fdt,
                                                            double* __restrict__ hfs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double hu = (x[N - 1] - x[0]) / (double)(N - 1);
    for (int b = blockIdx.y; b < B; b += gridDim.y) {      // grid.y is capped at 65535
        const double m = mu2[b];
        double frPrev = 0.0, fl = 0.0, hf = 0.0;
        if (i > 0) {
            const double xl = x[i - 1], xr = x[i];
            const double h = nonuniform ? xr - xl : hu;
            const double f1 = 0.02 * exp(m * (GP_A * xl + GP_B * xr));
            const double f2 = 0.02 * exp(m * (GP_B * xl + GP_A * xr));
            frPrev = (f1 * GP_B + f2 * GP_A) * (0.5 * h);
        }
        if (i < N - 1) {
            const double xl = x[i], xr = x[i + 1];
            const double h = nonuniform ? xr - xl : hu;
            const double f1 = 0.02 * exp(m * (GP_A * xl + GP_B * xr));
            const double f2 = 0.02 * exp(m * (GP_B * xl + GP_A * xr));
            fl = (f1 * GP_A + f2 * GP_B) * (0.5 * h);
            hf = h * (f1 + f2);
        }
        fdt[(size_t)b * N + i] = dt * (frPrev + fl);
        hfs[(size_t)b * N + i] = hf;
    }
}

// g = M u^n + dt F      reference: `M @ U[:, n] + At*F` of :683
__global__ __launch_bounds__(256) void mass_rhs_kernel(const double* __restrict__ x,
                                                       const double* __restrict__ un,
                                                       const double* __restrict__ fdt, int N, int B,
                                                       int nonuniform, double* __restrict__ g)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double h6 = (x[N - 1] - x[0]) / (double)(N - 1) / 6.0;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {          // grid.y is capped at 65535
        const double* u = un + (size_t)b * N;
        if (nonuniform) {
            double v = 0.0;
            if (i > 0) v = (x[i] - x[i - 1]) / 6.0 * __builtin_fma(2.0, u[i], u[i - 1]);
            if (i < N - 1) v = __builtin_fma((x[i + 1] - x[i]) / 6.0, __builtin_fma(2.0, u[i], u[i + 1]), v);
            g[(size_t)b * N + i] = v + fdt[(size_t)b * N + i];
            continue;
        }
        double acc;
        if (i == 0)
            acc = __builtin_fma(2.0, u[0], u[1]);
        else if (i == N - 1)
            acc = __builtin_fma(2.0, u[i], u[i - 1]);
        else
            acc = __builtin_fma(4.0, u[i], u[i - 1]) + u[i + 1];
        g[(size_t)b * N + i] = __builtin_fma(h6, acc, fdt[(size_t)b * N + i]);
    }
}

// ------------------------------------------------------------------------------------
// bg_rom_reduce: fused assembly + projection.
//   workgroup = 4 wavefronts; wave w, lane (g = lane>>4, c = lane&15) owns the mesh rows
//   i = (4w + g)*S + s, s = 0..S-1 as the K index of v_mfma_f64_16x16x4_f64 (k = g), so the
//   rows i-1, i, i+1 needed by the tridiagonal apply Y = A W sit in the SAME lane's adjacent
//   fragment registers.  Fragment t of step s holds W[i][16 t + c].  With a shared basis
//   (w_stride == 0) the fragments are loaded once per workgroup and stay in registers
//   while the workgroup walks over its samples.
// ------------------------------------------------------------------------------------
struct ReduceArgs {
    const double* x;
    const double* W;         // [N][r] (shared) or [B][N][r]
    long long w_stride;      // 0 = shared
    const double* U;         // [B][N]  current iterate
    const double* G;         // [B][N]  M u^n + dt F
    const double* hfs;       // [B][N]
    const double* mu1;       // [B]
    const int32_t* active;   // [B] or null
    double* Ar;              // [B][r][r]
    double* br;              // [B][r]
    double* wtu;             // [B][r]  W^T u  (null = skip)
    const double* q_in;      // [B][r] or null: if given, u = W q is formed here and stored to Uout
    double* Uout;            // [B][N] (only with q_in)
    double dt, E;
    int N, B, r, proj, supg, lift_only, nonuniform;
    int w_frag;              // 1: W is in the fragment-major layout of bg_quad_tangent (rom_reduce4 only)
    int w_colmajor;          // 1: W is [r][N] per sample (BG_OPT_W_COLMAJOR)
    const int32_t* w_index;  // [B] or null: sample b uses block w_index[b] of W (blocks w_stride apart)
};

template <int S, int NT, int PROJ>
__global__ __launch_bounds__(256, 1) void rom_reduce_kernel(ReduceArgs a)
{
    constexpr int NPAD = 16 * S;
    constexpr int RP = 16 * NT;             // padded reduced dimension
    __shared__ double s_u[NPAD + 2];        // u with one halo entry on each side
    __shared__ double s_coef[NPAD][4];      // lo, di, up, rhs(= -R) per row
    __shared__ double s_red[4][RP][RP + 1]; // per-wave partial Ar | br
    __shared__ double s_q[4][4][RP];        // per-wave, per-group partial W^T u

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int N = a.N, r = a.r;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const MeshConst mc = make_mesh_const(h, a.dt, a.E, a.supg);
    const int rowbase = (4 * w + g) * S;

    double frag[NT][S + 2];                 // W[rowbase + s - 1][16 t + c], s = 0..S+1
    bool have_frags = false;
    int last_block = -1;

    for (int smp = blockIdx.x; smp < a.B; smp += gridDim.x) {
        if (a.active && a.active[smp] == 0) continue;        // workgroup-uniform
        // ---- basis / tangent fragments (kept across samples while the block of W stays the same) -------
        const int wblock = a.w_stride == 0 ? 0 : (a.w_index ? a.w_index[smp] : smp);
        if (!have_frags || wblock != last_block) {
            last_block = wblock;
            const double* Wp = a.W + (size_t)wblock * (size_t)a.w_stride;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = 16 * t + c;
#pragma unroll
                for (int s = 0; s < S + 2; ++s) {
                    const int i = rowbase + s - 1;
                    frag[t][s] = (i >= 0 && i < N && col < r)
                                     ? (a.w_colmajor ? Wp[(size_t)col * N + i] : Wp[(size_t)i * r + col]) : 0.0;
                }
            }
            have_frags = true;
        }
        // ---- stage u (from HBM, or lifted u = W q from the register-resident basis) ----------
        if (a.q_in) {
            double qv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = 16 * t + c;
                qv[t] = col < r ? a.q_in[(size_t)smp * r + col] : 0.0;
            }
            if (tid == 0) { s_u[0] = 0.0; s_u[NPAD + 1] = 0.0; }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double p = 0.0;
#pragma unroll
                for (int t = 0; t < NT; ++t) p = __builtin_fma(frag[t][s + 1], qv[t], p);
                p += dpp_mov<0x111>(p);            // sum over the 16 lanes of the DPP row
                p += dpp_mov<0x112>(p);
                p += dpp_mov<0x114>(p);
                p += dpp_mov<0x118>(p);
                if (c == 15) {
                    const int i = rowbase + s;
                    s_u[i + 1] = p;
                    if (i < N) a.Uout[(size_t)smp * N + i] = p;
                }
            }
        } else {
            const double* up_ = a.U + (size_t)smp * N;
            for (int i = tid; i < NPAD + 2; i += 256) {
                const int gi = i - 1;
                s_u[i] = (gi >= 0 && gi < N) ? up_[gi] : 0.0;
            }
        }
        __syncthreads();
        if (a.lift_only) continue;           // workgroup-uniform
        const double mu1 = a.mu1[smp];
        for (int i = tid; i < NPAD; i += 256) {
            double lo, di, up, R;
            const bool in = i < N;
            rom_assemble_row(i, N, s_u[i], s_u[i + 1], s_u[i + 2], in ? a.G[(size_t)smp * N + i] : 0.0,
                             (in && i > 0) ? a.hfs[(size_t)smp * N + i - 1] : 0.0,
                             (in && i < N - 1) ? a.hfs[(size_t)smp * N + i] : 0.0, mu1, mc, a.nonuniform, a.x, a.dt, a.E,
                             lo, di, up, R);
            s_coef[i][0] = lo; s_coef[i][1] = di; s_coef[i][2] = up; s_coef[i][3] = R;
        }
        __syncthreads();
        // ---- MFMA contraction over this wave's rows -------------------------------------------
        f64x4 acc[NT][NT];
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb) acc[ta][tb] = f64x4{0.0, 0.0, 0.0, 0.0};
        double qp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) qp[t] = 0.0;
        const bool rcol_lane = (c == r - 16 * (NT - 1));      // lane that carries column r (= R)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = rowbase + s;
            const double lo = s_coef[i][0], di = s_coef[i][1], up = s_coef[i][2], R = s_coef[i][3];
            const double ui = s_u[i + 1];
            double Y[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double y = lo * frag[t][s];
                y = __builtin_fma(di, frag[t][s + 1], y);
                y = __builtin_fma(up, frag[t][s + 2], y);
                Y[t] = y;
                qp[t] = __builtin_fma(frag[t][s + 1], ui, qp[t]);
            }
            double Bf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) Bf[t] = Y[t];
            Bf[NT - 1] = rcol_lane ? R : Bf[NT - 1];
            if constexpr (PROJ == BG_PROJ_GALERKIN) {
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NT; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[ta][s + 1], Bf[tb], acc[ta][tb], 0, 0, 0);
            } else {
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int tb = ta; tb < NT; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[ta], Bf[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        // ---- cross-wave reduction through LDS, then write Ar | br | W^T u ----------------------
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                for (int k = 0; k < 4; ++k) s_red[w][16 * ta + g + 4 * k][16 * tb + c] = acc[ta][tb][k];
#pragma unroll
        for (int t = 0; t < NT; ++t) s_q[w][g][16 * t + c] = qp[t];
        __syncthreads();
        constexpr bool sym = PROJ != BG_PROJ_GALERKIN;
        for (int e = tid; e < r * (r + 1); e += 256) {
            const int row = e / (r + 1), col = e % (r + 1);
            int rr = row, cc = col;
            if (sym && col < r && (row >> 4) > (col >> 4)) { rr = col; cc = row; }   // mirror lower tiles
            const double v = (s_red[0][rr][cc] + s_red[1][rr][cc]) + (s_red[2][rr][cc] + s_red[3][rr][cc]);
            if (col < r)
                a.Ar[((size_t)smp * r + row) * r + col] = v;
            else
                a.br[(size_t)smp * r + row] = v;
        }
        if (a.wtu) {
            for (int j = tid; j < r; j += 256) {
                double v = 0.0;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) v += s_q[ww][gg][j];
                a.wtu[(size_t)smp * r + j] = v;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// rom_reduce4_kernel: the same fused assembly + projection on v_mfma_f64_4x4x4_4b_f64.
//   Measured on gfx950 (tools/mfma_bench.hip): the 4-block 4x4x4 form issues every ~16 cycles
//   (512 flop) = 31 flop/clk/SIMD, twice the rate of the 16x16x4 form (128 cycles, 2048 flop),
//   and its 4-wide blocks waste no padding at r = 40.  Lane map found with one-hot operands
//   (tools/mfma_layout_probe.hip, profiles/r01_mfma_f64_4x4x4_lane_map.txt):
//       A: lane = 16 k + 4 blk + i      B: lane = 16 k + 4 blk + j      D: lane = 16 i + 4 blk + j
//   The four blocks of one instruction work on the SAME pair (a, b) of 4-column blocks of the
//   output but on different mesh rows (K is split over k AND blk): "owner" o = 16 wave + (lane>>2)
//   owns the S consecutive rows [o S, o S + S), lane t = lane & 3 holds W[row][4 c + t] for every
//   column block c, so neighbouring rows are again the same lane's adjacent registers.  One extra
//   B block [R, u, 0, 0] yields br and W^T u.  The four block partials of every pair are summed at
//   the end with two DPP row shifts, the four waves through LDS.
// ------------------------------------------------------------------------------------
template <int NB, int PROJ, bool WTU>
struct Pairs4 {
    static constexpr int main_pairs = (PROJ == BG_PROJ_GALERKIN) ? NB * (NB + 1) : NB * (NB + 1) / 2 + NB;
    static constexpr int total = main_pairs + ((PROJ != BG_PROJ_GALERKIN && WTU) ? NB : 0);
};

template <int S, int NB, int PROJ, bool WTU>
__global__ __launch_bounds__(256, 1) void rom_reduce4_kernel(ReduceArgs a)
{
    constexpr int NPAD = 64 * S;
    constexpr int NPAIR = Pairs4<NB, PROJ, WTU>::total;
    constexpr int RW = 4 * NB;                   // padded reduced dimension
    constexpr bool GAL = PROJ == BG_PROJ_GALERKIN;
    // u, G, hfs of the current sample; double-buffered so that the NEXT sample's rows can stream in
    // with global_load_lds (no VGPRs, no wait) while this sample's MFMAs run.  u sits at offset 2
    // (16-byte aligned for the LDS DMA) with one zero halo entry on each side.
    __shared__ __attribute__((aligned(16))) double s_u[2][NPAD + 4];
    __shared__ __attribute__((aligned(16))) double s_g[2][NPAD];
    __shared__ __attribute__((aligned(16))) double s_h[2][NPAD];
    __shared__ __attribute__((aligned(16))) double s_qp[2][RW + 24];   // prefetched q (lifted mode), dword DMA
    __shared__ double s_coef[NPAD][4];
    __shared__ double s_red[5][RW][RW + 4];      // per-wave  Ar | [br, W^T u (Galerkin), 0, 0]; [4] = dump for idle lanes
    __shared__ double s_wtu[5][RW];              // per-wave  W^T u (LSPG); [4] = dump

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t = lane & 3, owner = 16 * w + (lane >> 2);
    const int N = a.N, r = a.r;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const MeshConst mc = make_mesh_const(h, a.dt, a.E, a.supg);
    const int rowbase = owner * S;

    double frag[NB][S + 2];                      // W[rowbase + s - 1][4 c + t]
    bool have_frags = false;

    for (int i = tid; i < 2 * (NPAD + 4); i += 256) (&s_u[0][0])[i] = 0.0;   // halos and padded rows stay finite
    for (int i = tid; i < 2 * NPAD; i += 256) { (&s_g[0][0])[i] = 0.0; (&s_h[0][0])[i] = 0.0; }
    __syncthreads();
    // LDS DMA needs 16-byte aligned, fully in-range source chunks: even N (row starts stay aligned)
    const bool pref = (N % 2 == 0) && !a.lift_only;
    auto next_active = [&](int from) {
        int n2 = from;
        while (n2 < a.B && a.active && a.active[n2] == 0) n2 += gridDim.x;
        return n2;
    };
    auto prefetch = [&](int smp_n, int buf_n) {
        const int idx = w * 128 + lane * 2;              // each wave streams 1 KiB per array
        if (idx < NPAD && idx + 1 < N) {
            const size_t o = (size_t)smp_n * N + idx;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.G + o), (lds_void_t*)(&s_g[buf_n][w * 128]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.hfs + o), (lds_void_t*)(&s_h[buf_n][w * 128]), 16, 0, 0);
            if (!a.q_in)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.U + o), (lds_void_t*)(&s_u[buf_n][2 + w * 128]), 16, 0, 0);
        }
        if (a.q_in && w == 0) {                              // q[smp_n][0..r): 2r dwords, 64 per instruction
            const float* qsrc = reinterpret_cast<const float*>(a.q_in + (size_t)smp_n * r);
            if (lane < 2 * r)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(qsrc + lane), (lds_void_t*)(&s_qp[buf_n][0]), 4, 0, 0);
            if (lane + 64 < 2 * r)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(qsrc + 64 + lane), (lds_void_t*)(&s_qp[buf_n][32]), 4, 0, 0);
        }
    };
    int smp = next_active(blockIdx.x);
    int buf = 0;
    if (pref && smp < a.B) prefetch(smp, 0);

    int last_block = -1;
    while (smp < a.B) {
        const int nxt = next_active(smp + gridDim.x);
        const int wblock = a.w_stride == 0 ? 0 : (a.w_index ? a.w_index[smp] : smp);
        if (!have_frags || wblock != last_block) {     // fragments stay in registers while the block of W stays the same
            last_block = wblock;
            const double* Wp = a.W + (size_t)wblock * (size_t)a.w_stride;
            if (a.w_frag) {
                // fragment-major: element (row o*S + s, col 4c + t) at ((c*S + s)*64 + o)*4 + t:
                // every (c, s) is one coalesced 2 KB read for the workgroup; halos come from owners o -+ 1
#pragma unroll
                for (int c = 0; c < NB; ++c) {
#pragma unroll
                    for (int s = 0; s < S + 2; ++s) {
                        const int i = rowbase + s - 1;
                        const int o = (s == 0) ? owner - 1 : ((s == S + 1) ? owner + 1 : owner);
                        const int ss = (s == 0) ? S - 1 : ((s == S + 1) ? 0 : s - 1);
                        frag[c][s] = (i >= 0 && i < N) ? Wp[((size_t)(c * S + ss) * 64 + o) * 4 + t] : 0.0;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const int col = 4 * c + t;
#pragma unroll
                    for (int s = 0; s < S + 2; ++s) {
                        const int i = rowbase + s - 1;
                        frag[c][s] = (i >= 0 && i < N && col < r)
                                         ? (a.w_colmajor ? Wp[(size_t)col * N + i] : Wp[(size_t)i * r + col]) : 0.0;
                    }
                }
            }
            have_frags = true;
        }
        // ---- stage u (from HBM, or lifted u = W q from the register-resident basis) ----------
        if (a.q_in) {
            double qv[NB];
            if (pref) {                                          // q arrived by LDS DMA one sample ago
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const int col = 4 * c + t;
                qv[c] = col < r ? (pref ? s_qp[buf][col] : a.q_in[(size_t)smp * r + col]) : 0.0;
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double p = 0.0;
#pragma unroll
                for (int c = 0; c < NB; ++c) p = __builtin_fma(frag[c][s + 1], qv[c], p);
                p += dpp_mov<0xB1>(p);             // quad_perm [1,0,3,2]: sum over the four t lanes
                p += dpp_mov<0x4E>(p);             // quad_perm [2,3,0,1]
                if (t == 0) {
                    const int i = rowbase + s;
                    s_u[buf][i + 2] = (i < N) ? p : 0.0;
                    if (i < N) a.Uout[(size_t)smp * N + i] = p;
                }
            }
        } else if (!pref) {
            const double* up_ = a.U + (size_t)smp * N;
            for (int i = tid; i < NPAD; i += 256) s_u[buf][i + 2] = (i < N) ? up_[i] : 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this sample's LDS DMA has landed (every wave waits for its own)
        __syncthreads();
        if (a.lift_only) { smp = nxt; continue; }                // workgroup-uniform
        // ---- assembly into LDS (same arithmetic as rom_reduce_kernel) -----------------------
        const double mu1 = a.mu1[smp];
        for (int i = tid; i < NPAD; i += 256) {
            double lo, di, up, R;
            const bool in = i < N;
            const double gi = in ? (pref ? s_g[buf][i] : a.G[(size_t)smp * N + i]) : 0.0;
            const double hL = (in && i > 0) ? (pref ? s_h[buf][i - 1] : a.hfs[(size_t)smp * N + i - 1]) : 0.0;
            const double hR = (in && i < N - 1) ? (pref ? s_h[buf][i] : a.hfs[(size_t)smp * N + i]) : 0.0;
            rom_assemble_row(i, N, s_u[buf][i + 1], s_u[buf][i + 2], (i + 1 < N) ? s_u[buf][i + 3] : 0.0, gi, hL, hR, mu1, mc,
                             a.nonuniform, a.x, a.dt, a.E, lo, di, up, R);
            s_coef[i][0] = lo; s_coef[i][1] = di; s_coef[i][2] = up; s_coef[i][3] = R;
        }
        __syncthreads();
        if (pref && nxt < a.B) prefetch(nxt, buf ^ 1);           // streams in under the MFMA phase
        // ---- MFMA contraction: every step feeds 16 mesh rows per wave ---------------------------
        double acc[NPAIR];
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) acc[p] = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = rowbase + s;
            const double lo = s_coef[i][0], di = s_coef[i][1], up = s_coef[i][2], R = s_coef[i][3];
            const double ui = s_u[buf][i + 2];
            double Y[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                double y = lo * frag[c][s];
                y = __builtin_fma(di, frag[c][s + 1], y);
                y = __builtin_fma(up, frag[c][s + 2], y);
                Y[c] = y;
            }
            const double X = (t == 0) ? R : ((t == 1) ? ui : 0.0);      // extra B block [R, u, 0, 0]
            int p = 0;
            if constexpr (GAL) {
#pragma unroll
                for (int ca = 0; ca < NB; ++ca) {
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb, ++p)
                        acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(frag[ca][s + 1], Y[cb], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(frag[ca][s + 1], X, acc[p], 0, 0, 0);
                    ++p;
                }
            } else {
#pragma unroll
                for (int ca = 0; ca < NB; ++ca) {
#pragma unroll
                    for (int cb = ca; cb < NB; ++cb, ++p)
                        acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(Y[ca], Y[cb], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(Y[ca], X, acc[p], 0, 0, 0);
                    ++p;
                }
                if constexpr (WTU) {
#pragma unroll
                    for (int ca = 0; ca < NB; ++ca, ++p)
                        acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(frag[ca][s + 1], X, acc[p], 0, 0, 0);
                }
            }
            // Galerkin (110 accumulators): nothing moves across a row step -- otherwise the scheduler hoists later steps' operand
            // loads and the register allocator answers with more scratch reloads inside the MFMA loop (412 -> 292 bytes per
            // lane, 290 -> 266 us per 4096 samples; the LSPG forms are 3 % faster without it)
            if constexpr (GAL) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sum the four block partials of every pair (lanes differing in bits 2..3) -----------
        {
            const int oi = lane >> 4, oj = lane & 3;
            const bool writer = ((lane >> 2) & 3) == 3;
            const int wslot = writer ? w : 4;            // idle lanes store into the dump slab: no branches
            int p = 0;
#pragma unroll
            for (int ca = 0; ca < NB; ++ca) {
#pragma unroll
                for (int cb = (GAL ? 0 : ca); cb <= NB; ++cb, ++p) {
                    double v = acc[p];
                    v += dpp_mov<0x114>(v);          // row_shr:4
                    v += dpp_mov<0x118>(v);          // row_shr:8 -> lanes with blk == 3 hold the sum
                    s_red[wslot][4 * ca + oi][4 * cb + oj] = v;
                }
            }
            if constexpr (!GAL && WTU) {
#pragma unroll
                for (int ca = 0; ca < NB; ++ca, ++p) {
                    double v = acc[p];
                    v += dpp_mov<0x114>(v);
                    v += dpp_mov<0x118>(v);
                    s_wtu[(writer && oj == 1) ? w : 4][4 * ca + oi] = v;
                }
            }
        }
        __syncthreads();
        {   // thread (col = tid & 63, row phase = tid >> 6): no integer division, coalesced row stores
            const int col = tid & 63;
            if (col <= r) {
                for (int row = tid >> 6; row < r; row += 4) {
                    int rr = row, cc = (col < r) ? col : RW;             // br sits in column RW
                    if (!GAL && col < r && (row >> 2) > (col >> 2)) { rr = col; cc = row; }   // mirror lower blocks
                    const double v = (s_red[0][rr][cc] + s_red[1][rr][cc]) + (s_red[2][rr][cc] + s_red[3][rr][cc]);
                    if (col < r)
                        a.Ar[((size_t)smp * r + row) * r + col] = v;
                    else
                        a.br[(size_t)smp * r + row] = v;
                }
            }
        }
        if (a.wtu) {
            for (int j = tid; j < r; j += 256) {
                double v;
                if constexpr (GAL)
                    v = (s_red[0][j][RW + 1] + s_red[1][j][RW + 1]) + (s_red[2][j][RW + 1] + s_red[3][j][RW + 1]);
                else if constexpr (WTU)
                    v = (s_wtu[0][j] + s_wtu[1][j]) + (s_wtu[2][j] + s_wtu[3][j]);
                else
                    v = 0.0;
                a.wtu[(size_t)smp * r + j] = v;
            }
        }
        __syncthreads();
        smp = nxt;
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------
// quad_tangent_kernel: T[b] = Phi + H3 . q[b]  (reference tangent(), FEM/fem_burgers.py:1120-1123,
// with H3[i][a][c] = H[i][pair(a,c)] (1 + delta_ac) folding get_dQ_dq :292-312), written straight
// into the fragment-major layout rom_reduce4_kernel reads.  Across the samples this is a GEMM per mesh row,
// T_i (n x B) = H3_i (n x n) . Q (n x B), and it runs on v_mfma_f64_4x4x4_4b: workgroup = one (column block c, row
// slot s) and a chunk of 64 samples; MFMA block = one mesh row (owner), A = H3[row][4 columns][4 k] stationary in
// registers (NKC doubles per lane and owner group), B = q[4 samples][4 k] from the chunk's copy in LDS (one 512-byte read
// per wave, 4 samples and 4 k -- the first version fed q to VALU FMAs through scalar loads, whose latency it could not hide:
// 69 us per launch at config 3 against a 21 us arithmetic bound), D = [4 columns][4 samples] per owner, which is 128
// contiguous bytes of a sample's fragment block per 16 lanes: stored straight from the accumulators.
// ------------------------------------------------------------------------------------
template <int S, int NKC>
__global__ __launch_bounds__(256) void quad_tangent_kernel(const double* __restrict__ Phi, const double* __restrict__ H3p,
                                                           const double* __restrict__ qp,
                                                           const int32_t* __restrict__ active, double* __restrict__ Wf,
                                                           int N, int B, int n, int NB, int chunk)
{
    // H3p is [N][n][NP] and qp is [B][NP], NP = 4 NKC, both zero-padded in their last dimension: no tail tests
    constexpr int NP = 4 * NKC;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int* s_on = reinterpret_cast<int*>(s_dyn);                                  // [chunk]
    double (*s_q)[NP] = reinterpret_cast<double (*)[NP]>(s_dyn + 4 * chunk);    // [chunk][NP]; chunk is a multiple of 4
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = blockIdx.x / S, s = blockIdx.x % S;
    const int b0 = blockIdx.y * chunk, nb = min(chunk, B - b0);
    for (int e = tid; e < chunk * NP; e += 256) {
        const int sb = e / NP, kk = e - sb * NP;
        s_q[sb][kk] = sb < nb ? qp[(size_t)(b0 + sb) * NP + kk] : 0.0;
    }
    for (int e = tid; e < chunk; e += 256) s_on[e] = (e < nb && (!active || active[b0 + e] != 0)) ? 1 : 0;
    // A operand lane: 16 k + 4 blk + i, i = column within the block; owner group og: owners 16 w + 4 og + blk
    const int ak = lane >> 4, ablk = (lane >> 2) & 3, at = lane & 3;
    double a[4][NKC];
#pragma unroll
    for (int og = 0; og < 4; ++og) {
        const int i = (16 * w + 4 * og + ablk) * S + s, col = 4 * c + at;
        const bool live = i < N && col < n;
        const double* hp = H3p + ((size_t)(live ? i : 0) * n + (live ? col : 0)) * NP + ak;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const double v = hp[4 * kc];                               // always a valid address: no branch per load
            a[og][kc] = live ? v : 0.0;
        }
    }
    // result lane: 16 i + 4 blk + j holds T[owner blk][column i][sample j]
    const int dt = lane >> 4, dblk = (lane >> 2) & 3, dj = lane & 3;
    double phi[4];
#pragma unroll
    for (int og = 0; og < 4; ++og) {
        const int i = (16 * w + 4 * og + dblk) * S + s, col = 4 * c + dt;
        phi[og] = (i < N && col < n) ? Phi[(size_t)i * n + col] : 0.0;
    }
    const size_t per_sample = (size_t)NB * S * 256;
    const size_t off = ((size_t)(c * S + s) * 64 + 16 * w + dblk) * 4 + dt;      // + 16 og for owner group og
    __syncthreads();
    for (int sg = 0; sg < chunk / 4; ++sg) {
        const int4 on = *reinterpret_cast<const int4*>(&s_on[4 * sg]);
        if (!(on.x | on.y | on.z | on.w)) continue;        // workgroup-uniform: four converged (or absent) samples
        double acc[4] = {phi[0], phi[1], phi[2], phi[3]};
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const double bq = s_q[4 * sg + at][4 * kc + ak];          // B operand lane: 16 k + 4 blk + j
#pragma unroll
            for (int og = 0; og < 4; ++og) acc[og] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[og][kc], bq, acc[og], 0, 0, 0);
        }
        if (s_on[4 * sg + dj]) {
            double* dst = Wf + (size_t)(b0 + 4 * sg + dj) * per_sample + off;
#pragma unroll
            for (int og = 0; og < 4; ++og) dst[16 * og] = acc[og];
        }
    }
}

// ------------------------------------------------------------------------------------
// quad_features_kernel: the left operand of the quadratic-manifold decode u = Phi q + H Q(q) (:1116-1118) as ONE matrix,
// feat[b] = [q[b] | Q(q[b])] with Q = the unique products q_i q_j, j >= i, in get_sym's order (:263-273), so that the
// decode is one GEMM against [Phi^T; H^T] instead of two gathers, a product, two GEMMs and an add.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quad_features_kernel(const double* __restrict__ q, const int32_t* __restrict__ pi,
                                                            const int32_t* __restrict__ pj, double* __restrict__ feat, int B,
                                                            int n, int k)
{
    const int w = n + k;
    const long long total = (long long)B * w;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int b = (int)(e / w), c = (int)(e % w);
        const double* qb = q + (size_t)b * n;
        feat[e] = (c < n) ? qb[c] : qb[pi[c - n]] * qb[pj[c - n]];
    }
}

// ------------------------------------------------------------------------------------
// bg_lu_solve: x = solve(A, sign * b), partial pivoting, one wavefront per system.
//   lane i holds row i of [A | b]; rows are never moved: the pivot of step k is the
//   not-yet-used lane with the largest |a[k]| (LAPACK gesv's choice up to ties in the
//   top 32 bits), its row is broadcast with v_readlane.  reference: np.linalg.solve :767.
// ------------------------------------------------------------------------------------
struct LuArgs {
    const double* A;      // [B][n][n]
    const double* b;      // [B][n]
    const int32_t* active;
    double* x;            // [B][n]
    int32_t* info;        // [B]  0 ok, k+1 = zero pivot at step k
    double sign;
    int n, B;
    // fused iteration update (mode != 0): see bg_lu_solve_update
    int mode, max_it;
    double tol;
    const double* wtu;    // [B][n]  (mode 1)
    double* q;            // [B][n]  in/out
    int32_t* active_io;   // [B]
    int32_t* iters;       // [B]
    int32_t* flags;       // [B]
    int32_t* counter;     // [2 * BG_COUNTER_SLOTS * BG_COUNTER_STRIDE]  partial counts: still active | singular (see burgers_hip.h)
};

template <int NMAX>
__global__ __launch_bounds__(256) void lu_solve_kernel(LuArgs a)
{
    const int lane = threadIdx.x & 63;
    const int sys = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sys >= a.B) return;
    if (a.active && a.active[sys] == 0) return;
    if (a.mode != 0 && a.active_io[sys] == 0) return;
    const int n = a.n;
    double row[NMAX + 1];
    const double* Ap = a.A + (size_t)sys * n * n;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
        row[j] = (lane < n && j < n) ? Ap[(size_t)lane * n + j] : ((lane == j) ? 1.0 : 0.0);
    row[NMAX] = (lane < n) ? a.sign * a.b[(size_t)sys * n + lane] : 0.0;

    int info;
    const double xout = lu_pivoted_wave<NMAX>(row, lane, info);
    if (lane < n) a.x[(size_t)sys * n + lane] = xout;
    if (lane == 0 && a.info && info) a.info[sys] = info;
    if (a.mode != 0) {
        if (lane == 0 && info) atomicAdd(a.counter + (BG_COUNTER_SLOTS + (sys & (BG_COUNTER_SLOTS - 1))) * BG_COUNTER_STRIDE, 1);
        // reference updates: POD  q = Phi^T U0 + dq, err = |dq|/|q|                    (:770-776)
        //                    quad q += dq, rel = |dq|/max(1e-14,|q|), stop if rel<tol   (:1161-1169)
        //                    ANN  q_p += dq, err = |dq|/(|q_p|+1e-14)                   (:1237-1244)
        double base = 0.0;
        if (lane < n) base = (a.mode == 1) ? a.wtu[(size_t)sys * n + lane] : a.q[(size_t)sys * n + lane];
        const double qn = (lane < n) ? base + xout : 0.0;
        if (lane < n) a.q[(size_t)sys * n + lane] = qn;
        double nd, nq;
        wave_sum2(xout * xout, qn * qn, nd, nq);
        nd = sqrt(nd); nq = sqrt(nq);
        const int k = a.iters[sys] + 1;
        bool more;
        double err;
        if (a.mode == 1) { err = nd / nq; more = (err > a.tol) && (k < a.max_it); }
        else if (a.mode == 2) { err = nd / fmax(1e-14, nq); more = !(err < a.tol) && (k < a.max_it); }
        else { err = nd / (nq + 1e-14); more = (err > a.tol) && (k < a.max_it); }
        if (lane == 0) {
            a.iters[sys] = k;
            a.active_io[sys] = more ? 1 : 0;
            int f = 0;
            if (!(err - err == 0.0)) f |= BG_FLAG_NONFINITE;
            if (k >= a.max_it && ((a.mode == 2) ? !(err < a.tol) : true)) f |= BG_FLAG_HIT_CAP;
            if (f) a.flags[sys] |= f;
            if (more) atomicAdd(a.counter + (sys & (BG_COUNTER_SLOTS - 1)) * BG_COUNTER_STRIDE, 1);   // one cache line per slot
        }
    }
}

template <typename F>
int dispatch_lu(int n, F&& f)
{
    if (n <= 8) return f(std::integral_constant<int, 8>{});
    if (n <= 16) return f(std::integral_constant<int, 16>{});
    if (n <= 24) return f(std::integral_constant<int, 24>{});
    if (n <= 32) return f(std::integral_constant<int, 32>{});
    if (n <= 40) return f(std::integral_constant<int, 40>{});
    if (n <= 48) return f(std::integral_constant<int, 48>{});
    if (n <= 64) return f(std::integral_constant<int, 64>{});
    return BG_ERR_UNSUPPORTED_R;
}

}  // namespace

extern "C" {

int bg_forcing_setup(int N, int B, const double* x, const double* mu2, double dt, int options, double* fdt,
                     double* hfs, void* stream)
{
    if (N < 2 || B < 0) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!x || !mu2 || !fdt || !hfs) return BG_ERR_BAD_ARG;
    hipLaunchKernelGGL(forcing_setup_kernel, dim3((N + 255) / 256, B < 65535 ? B : 65535), dim3(256), 0, (hipStream_t)stream, x, mu2, N,
                       B, dt, (options & BG_OPT_NONUNIFORM) ? 1 : 0, fdt, hfs);
    return check_launch();
}

int bg_mass_rhs(int N, int B, const double* x, const double* un, const double* fdt, int options, double* g,
                void* stream)
{
    if (N < 2 || B < 0) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!x || !un || !fdt || !g) return BG_ERR_BAD_ARG;
    hipLaunchKernelGGL(mass_rhs_kernel, dim3((N + 255) / 256, B < 65535 ? B : 65535), dim3(256), 0, (hipStream_t)stream, x, un, fdt, N, B,
                       (options & BG_OPT_NONUNIFORM) ? 1 : 0, g);
    return check_launch();
}

int bg_rom_max_n(void) { return 512; }
int bg_rom_max_r(void) { return 47; }

static int rom_reduce_impl(int N, int B, int r, int projection, const double* x, const double* W, long long w_stride,
                           const double* U, const double* G, const double* hfs, const double* mu1, double dt, double E,
                           int supg, const int32_t* active, double* Ar, double* br, double* wtu, const double* q_in,
                           double* Uout, int lift_only, void* stream, int w_frag = 0, const int32_t* w_index = nullptr)
{
    if (N < 2 || B < 0 || r < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (projection != BG_PROJ_GALERKIN && projection != BG_PROJ_LSPG) return BG_ERR_PROJECTION;
    if (N > 512) return BG_ERR_UNSUPPORTED_N;
    if (r > 47) return BG_ERR_UNSUPPORTED_R;
    if (B == 0) return BG_OK;
    if (!x || !W) return BG_ERR_BAD_ARG;
    if (!lift_only && (!G || !hfs || !mu1 || !Ar || !br)) return BG_ERR_BAD_ARG;
    if (q_in ? (!Uout || w_stride != 0) : !U) return BG_ERR_BAD_ARG;
    ReduceArgs a;
    a.x = x; a.W = W; a.w_stride = w_stride; a.U = U; a.G = G; a.hfs = hfs; a.mu1 = mu1; a.active = active;
    a.Ar = Ar; a.br = br; a.wtu = wtu; a.dt = dt; a.E = E; a.N = N; a.B = B; a.r = r; a.proj = projection;
    a.supg = supg & BG_OPT_SUPG; a.nonuniform = (supg & BG_OPT_NONUNIFORM) ? 1 : 0;
    a.q_in = q_in; a.Uout = Uout; a.lift_only = lift_only; a.w_frag = w_frag;
    a.w_colmajor = (supg & BG_OPT_W_COLMAJOR) ? 1 : 0;
    a.w_index = w_index;
    const bool force16 = (supg & BG_OPT_MFMA_16X16) != 0;
    if (w_frag && (r > 40 || force16)) return BG_ERR_UNSUPPORTED_R;
    const int cus = device_cu_count();
    const int grid = B < cus ? B : cus;
    hipStream_t st = (hipStream_t)stream;
    // fast path: v_mfma_f64_4x4x4_4b kernel, r <= 40 (accumulators and fragments must fit the register file)
    if (r <= 40 && !force16) {
        const int nb = r <= 8 ? 2 : (r <= 16 ? 4 : (r <= 24 ? 6 : (r <= 32 ? 8 : 10)));
        const bool want_wtu = wtu != nullptr;
        const bool gal = projection == BG_PROJ_GALERKIN;
        const int s4 = N <= 256 ? 4 : 8;
#define BG_LAUNCH_R4(SV, NBV)                                                                                      \
    do {                                                                                                           \
        if (gal)                                                                                                   \
            hipLaunchKernelGGL((rom_reduce4_kernel<SV, NBV, BG_PROJ_GALERKIN, true>), dim3(grid), dim3(256), 0, st, a); \
        else if (want_wtu)                                                                                         \
            hipLaunchKernelGGL((rom_reduce4_kernel<SV, NBV, BG_PROJ_LSPG, true>), dim3(grid), dim3(256), 0, st, a);    \
        else                                                                                                       \
            hipLaunchKernelGGL((rom_reduce4_kernel<SV, NBV, BG_PROJ_LSPG, false>), dim3(grid), dim3(256), 0, st, a);   \
    } while (0)
        switch (s4 * 100 + nb) {
            case 402: BG_LAUNCH_R4(4, 2); break;
            case 404: BG_LAUNCH_R4(4, 4); break;
            case 406: BG_LAUNCH_R4(4, 6); break;
            case 408: BG_LAUNCH_R4(4, 8); break;
            case 410: BG_LAUNCH_R4(4, 10); break;
            case 802: BG_LAUNCH_R4(8, 2); break;
            case 804: BG_LAUNCH_R4(8, 4); break;
            case 806: BG_LAUNCH_R4(8, 6); break;
            case 808: BG_LAUNCH_R4(8, 8); break;
            case 810: BG_LAUNCH_R4(8, 10); break;
            default: return BG_ERR_UNSUPPORTED_R;
        }
#undef BG_LAUNCH_R4
        return check_launch();
    }
    const int S = N <= 128 ? 8 : (N <= 256 ? 16 : 32);
    const int NT = (r + 1 + 15) / 16;     // room for the extra column that carries R
#define BG_LAUNCH_REDUCE(SV, NTV)                                                                              \
    do {                                                                                                       \
        if (projection == BG_PROJ_GALERKIN)                                                                    \
            hipLaunchKernelGGL((rom_reduce_kernel<SV, NTV, BG_PROJ_GALERKIN>), dim3(grid), dim3(256), 0, st, a); \
        else                                                                                                   \
            hipLaunchKernelGGL((rom_reduce_kernel<SV, NTV, BG_PROJ_LSPG>), dim3(grid), dim3(256), 0, st, a);     \
    } while (0)
    switch (S * 10 + NT) {
        case 81: BG_LAUNCH_REDUCE(8, 1); break;
        case 82: BG_LAUNCH_REDUCE(8, 2); break;
        case 83: BG_LAUNCH_REDUCE(8, 3); break;
        case 161: BG_LAUNCH_REDUCE(16, 1); break;
        case 162: BG_LAUNCH_REDUCE(16, 2); break;
        case 163: BG_LAUNCH_REDUCE(16, 3); break;
        case 321: BG_LAUNCH_REDUCE(32, 1); break;
        case 322: BG_LAUNCH_REDUCE(32, 2); break;
        case 323: BG_LAUNCH_REDUCE(32, 3); break;
        default: return BG_ERR_UNSUPPORTED_R;
    }
#undef BG_LAUNCH_REDUCE
    return check_launch();
}

int bg_rom_reduce(int N, int B, int r, int projection, const double* x, const double* W, long long w_stride,
                  const double* U, const double* G, const double* hfs, const double* mu1, double dt, double E,
                  int supg, const int32_t* active, double* Ar, double* br, double* wtu, void* stream)
{
    return rom_reduce_impl(N, B, r, projection, x, W, w_stride, U, G, hfs, mu1, dt, E, supg, active, Ar, br, wtu,
                           nullptr, nullptr, 0, stream);
}

int bg_rom_reduce_indexed(int N, int B, int r, int projection, const double* x, const double* W, long long w_stride,
                          const int32_t* w_index, const double* U, const double* G, const double* hfs, const double* mu1,
                          double dt, double E, int supg, const int32_t* active, double* Ar, double* br, double* wtu,
                          void* stream)
{
    if (B > 0 && (!w_index || w_stride == 0)) return BG_ERR_BAD_ARG;
    return rom_reduce_impl(N, B, r, projection, x, W, w_stride, U, G, hfs, mu1, dt, E, supg, active, Ar, br, wtu,
                           nullptr, nullptr, 0, stream, 0, w_index);
}

int bg_rom_reduce_lifted(int N, int B, int r, int projection, const double* x, const double* Phi, const double* q,
                         double* U, const double* G, const double* hfs, const double* mu1, double dt, double E,
                         int supg, const int32_t* active, double* Ar, double* br, double* wtu, void* stream)
{
    if (B == 0) return BG_OK;
    if (!q || !U) return BG_ERR_BAD_ARG;
    return rom_reduce_impl(N, B, r, projection, x, Phi, 0, nullptr, G, hfs, mu1, dt, E, supg, active, Ar, br, wtu, q,
                           U, 0, stream);
}

static int frag_nb(int r) { return r <= 8 ? 2 : (r <= 16 ? 4 : (r <= 24 ? 6 : (r <= 32 ? 8 : 10))); }
static int frag_s(int N) { return N <= 256 ? 4 : 8; }

int bg_rom_frag_pad(int r) { return (r < 1 || r > 40) ? 0 : 4 * frag_nb(r); }

long long bg_rom_frag_elems(int N, int r)
{
    if (N < 2 || N > 512 || r < 1 || r > 40) return 0;
    return (long long)frag_nb(r) * frag_s(N) * 256;
}

int bg_quad_tangent(int N, int B, int n, const double* Phi, const double* H3, const double* q,
                    const int32_t* active, double* Wfrag, void* stream)
{
    if (N < 2 || B < 0 || n < 1) return BG_ERR_BAD_ARG;
    if (N > 512) return BG_ERR_UNSUPPORTED_N;
    if (n > 40) return BG_ERR_UNSUPPORTED_R;
    if (B == 0) return BG_OK;
    if (!Phi || !H3 || !q || !Wfrag) return BG_ERR_BAD_ARG;
    const int NB = frag_nb(n), S = frag_s(N);                       // 4 NB: padded last dimension of H3p / qp
    // Samples per workgroup: the kernel runs 4 workgroups per CU (registers), so the sample chunks are sized to give at
    // most 4 CUs' worth of workgroups -- ONE resident round (64-sample chunks at config 3 made 5 per CU: a second round
    // with a single workgroup per CU doubled the launch) -- within 32 KB of LDS for the chunk's copy of q.
    const int slots = 4 * device_cu_count();
    const int max_chunk = ((32768 / (32 * NB)) / 4) * 4;
    int nchunks = slots / (NB * S);
    if (nchunks < 1) nchunks = 1;
    int chunk = (((B + nchunks - 1) / nchunks) + 3) & ~3;
    if (chunk < 32) chunk = 32;
    if (chunk > max_chunk) chunk = max_chunk;
    const dim3 grid(NB * S, (B + chunk - 1) / chunk), block(256);
    const size_t lds = (size_t)chunk * (4 + 32 * NB);
    hipStream_t st = (hipStream_t)stream;
#define BG_QT(SV, NKCV) hipLaunchKernelGGL((quad_tangent_kernel<SV, NKCV>), grid, block, lds, st, Phi, H3, q, active, Wfrag, N, B, n, NB, chunk)
    switch (S * 100 + NB) {
        case 402: BG_QT(4, 2); break;
        case 404: BG_QT(4, 4); break;
        case 406: BG_QT(4, 6); break;
        case 408: BG_QT(4, 8); break;
        case 410: BG_QT(4, 10); break;
        case 802: BG_QT(8, 2); break;
        case 804: BG_QT(8, 4); break;
        case 806: BG_QT(8, 6); break;
        case 808: BG_QT(8, 8); break;
        case 810: BG_QT(8, 10); break;
        default: return BG_ERR_UNSUPPORTED_R;
    }
#undef BG_QT
    return check_launch();
}

int bg_quad_features(int B, int n, const double* q, const int32_t* pair_i, const int32_t* pair_j, double* feat, void* stream)
{
    if (B < 0 || n < 1) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!q || !pair_i || !pair_j || !feat) return BG_ERR_BAD_ARG;
    const int k = n * (n + 1) / 2;
    const long long total = (long long)B * (n + k);
    const int grid = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(quad_features_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, q, pair_i, pair_j, feat, B, n, k);
    return check_launch();
}

int bg_rom_reduce_frag(int N, int B, int r, int projection, const double* x, const double* Wfrag, const double* U,
                       const double* G, const double* hfs, const double* mu1, double dt, double E, int supg,
                       const int32_t* active, double* Ar, double* br, double* wtu, void* stream)
{
    const long long per = bg_rom_frag_elems(N, r);
    if (per == 0) return (r > 40) ? BG_ERR_UNSUPPORTED_R : BG_ERR_UNSUPPORTED_N;
    return rom_reduce_impl(N, B, r, projection, x, Wfrag, per, U, G, hfs, mu1, dt, E, supg, active, Ar, br, wtu,
                           nullptr, nullptr, 0, stream, 1);
}

int bg_rom_lift(int N, int B, int r, const double* x, const double* Phi, const double* q, const int32_t* active,
                double* U, void* stream)
{
    if (B == 0) return BG_OK;
    if (!q || !U) return BG_ERR_BAD_ARG;
    return rom_reduce_impl(N, B, r, BG_PROJ_GALERKIN, x, Phi, 0, nullptr, nullptr, nullptr, nullptr, 1.0, 0.0, 0,
                           active, nullptr, nullptr, nullptr, q, U, 1, stream);
}

int bg_lu_solve_update(int n, int B, const double* A, const double* b, int mode, const double* wtu, double* q,
                       double* dq, double tol, int max_it, int32_t* active, int32_t* iters, int32_t* flags,
                       int32_t* counter, int32_t* info, void* stream)
{
    if (n < 1 || B < 0 || mode < 1 || mode > 3) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!A || !b || !q || !dq || !active || !iters || !flags || !counter || (mode == 1 && !wtu)) return BG_ERR_BAD_ARG;
    LuArgs a{A, b, nullptr, dq, info, -1.0, n, B, mode, max_it, tol, wtu, q, active, iters, flags, counter};
    hipStream_t st = (hipStream_t)stream;
    return dispatch_lu(n, [&](auto nc) {
        constexpr int NMAX = decltype(nc)::value;
        hipLaunchKernelGGL((lu_solve_kernel<NMAX>), dim3((B + 3) / 4), dim3(256), 0, st, a);
        return check_launch();
    });
}

int bg_lu_solve(int n, int B, const double* A, const double* b, double sign, const int32_t* active, double* x,
                int32_t* info, void* stream)
{
    if (n < 1 || B < 0) return BG_ERR_BAD_ARG;
    if (B == 0) return BG_OK;
    if (!A || !b || !x) return BG_ERR_BAD_ARG;
    LuArgs a{A, b, active, x, info, sign, n, B, 0, 0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    return dispatch_lu(n, [&](auto nc) {
        constexpr int NMAX = decltype(nc)::value;
        hipLaunchKernelGGL((lu_solve_kernel<NMAX>), dim3((B + 3) / 4), dim3(256), 0, st, a);
        return check_launch();
    });
}

}  // extern "C"
