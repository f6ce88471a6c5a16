// quad_fused.hip -- the whole quadratic-manifold PROM time loop of FOUR samples on one compute unit (bg_quad_rom_run).
//
// Replaces FEMBurgers.pod_quadratic_manifold (reference FEM/fem_burgers.py:1081-1175) for a batch of samples: decoder
// u = Phi q + H Q(q) (:1116-1118), tangent T = Phi + H dQ/dq (:1120-1123), A(u), R(u) (:1133-1147), the reduced system
// Ar = (A T)^T (A T) | T^T A T, br = (A T)^T R | T^T R (:1152-1158), solve, q += dq, the stopping test (:1161-1169).
//
// Why four samples per workgroup.  The tangent of ONE sample is a matrix-vector product per mesh row, T_i = Phi_i + H3_i q
// with H3[i][a][c] = H[i][pair(a, c)] (1 + delta_ac): 6.5 MB of H3 (N = 512, n = 40) streamed per sample and iteration, far
// beyond what L2 delivers.  The batched path (rom.hip) therefore forms the tangents of 64 samples at a time in one kernel
// and projects them in another, and the N x n tangent of every sample makes a round trip through memory in between
// (168 MB written and read back per iteration at B = 1024: 17 x the algorithmic bytes).  Here a
// 256-thread workgroup owns four samples for ALL time steps and iterations:
//   * tangent on v_mfma_f64_4x4x4_4b: one instruction = 4 mesh rows (blocks) x 4 columns x 4 SAMPLES (the workgroup's
//     four), contracting 4 k; the A operand is H3 in a fragment-major copy (one coalesced 16-byte load per lane and
//     2 k-chunks), so H3 is streamed once per FOUR sample-iterations and the tiles go to LDS, 64 mesh rows at a time;
//   * the decode needs no second contraction: H Q(q) = 1/2 (H3 q) q, so  u = 1/2 (Phi q + T(q) q)  falls out of the
//     tangent rows while they are in LDS (40 FMAs per row instead of the 860 of Phi q + H Q(q));
//   * wave w owns sample w: it assembles A(u), R(u) for its sample's 64 rows, reads its tangent rows back from LDS in
//     the projection's operand layout, applies the tridiagonal A on the fly and keeps ALL accumulators of its sample's
//     reduced system in registers across the row slabs (one wave per SIMD: 512 registers) -- no cross-wave reduction,
//     no partial systems; at the end of a pass the four block partials of each accumulator are summed with two DPP
//     steps, the wave solves its 40 x 40 system (lane = row, Gauss-Jordan with partial pivoting: np.linalg.solve's pivot
//     choice, no back substitution), updates q and tests |dq|/|q|.
// Nothing but H3 / Phi reads and one history row per time step touches memory; the host is not in the loop.
// A pass = one sweep over the mesh at the current q: iteration k uses T(q_k), u(q_k); after the last update one more
// sweep without projection yields u(q_K) = U[:, m+1] (samples that converged earlier simply keep their q).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/burgers_hip.h"
#include "abi_common.hpp"
#include "rom_device.hpp"

namespace {

using namespace bg;

constexpr int QN = 40;                 // padded reduced dimension: column 10 t + c  <->  (lane index t, block c)
constexpr int QNB = 10;                // 4-column blocks
constexpr int QKC = 10;                // k chunks of 4 of the tangent contraction
constexpr int QG = 4;                  // samples per workgroup = waves
constexpr int QRS = 64;                // mesh rows projected per slab
constexpr int QTR = QRS + 8;           // tangent rows held per slab: [r0 - 4, r0 + 68), 18 row groups of 4
constexpr int QTS = 42;                // doubles per tangent row in LDS (16-byte aligned rows, banks spread)
constexpr int QTJ = QTR * QTS + 4;     // doubles per sample in s_T
constexpr int QSYS = QN + 1;           // row length of the reduced system parked in LDS: Ar | br
constexpr int QPAIRS = QNB * (QNB + 1) / 2;   // 4 x 4 blocks (a <= b) of the symmetric tangent tensor of a mesh row: 55
constexpr int QP2 = (QPAIRS + 1) / 2;         // 16-byte load slots per lane and row group: 28
constexpr int QRING = QTR / 4;                // row groups the LDS ring of tangent rows holds: 18
constexpr int qpair(int a, int b) { return a * QNB - a * (a - 1) / 2 + (b - a); }      // a <= b, row-major upper triangle
constexpr int qrow(int p) { int a = 0; while (a + 1 < QNB && qpair(a + 1, a + 1) <= p) ++a; return a; }
constexpr int qcol(int p) { return qrow(p) + (p - qpair(qrow(p), qrow(p))); }
#ifndef BG_QUAD_WINDOW
#define BG_QUAD_WINDOW 14
#endif
constexpr int QW = BG_QUAD_WINDOW;            // 16-byte slots of the tangent stream in flight per wave (divides 28)
static_assert(QP2 % QW == 0, "the ring of slots runs across row groups with static indices");
#ifdef BG_QUAD_TIMING                   // diagnostic builds (tools/time_quad_fused.py): shader clocks per phase instead of the counts
constexpr bool kQT = true;
#else
constexpr bool kQT = false;
#endif

struct QuadRunArgs {
    const double* x;        // [N]
    const double* PhiT;     // [40][NPAD]   Phi^T, zero padded
    const double* Phif;     // [NG][10][16] Phi[4 rg + blk][4 c + i] at [rg][c][4 i + blk]: accumulator seed of tangent tile c
    const double* H3f;      // [NG][28][64][2]  upper 4 x 4 blocks of the symmetric tangent tensor, A-operand order (see bg_quad_rom_run)
    const double* u0;       // [B][N]
    const double* mu1;      // [B]
    const double* mu2;      // [B]
    double* hist;           // [B][nsteps+1][N]
    int32_t* iters;         // [B][nsteps]
    int32_t* flags;         // [B]
    int32_t* info;          // [B]
    const int32_t* order;   // [B] or null: wave w of group g works on sample order[4 g + w] (the four share their passes)
    double dt, E, tol;
    int N, NPAD, NG, B, n, nsteps, max_it, nonuniform;
};

template <bool GAL>
struct QuadAcc {
    static constexpr int main_pairs = GAL ? QNB * QNB : QNB * (QNB + 1) / 2;
    static constexpr int total = main_pairs + QNB;          // + the [R] column: br
};

template <bool GAL>
__global__ __launch_bounds__(256, 1) void quad_fused_kernel(QuadRunArgs a)
{
    constexpr int NACC = QuadAcc<GAL>::total;
    __shared__ __attribute__((aligned(16))) double s_T[QG * QTJ];          // tangent rows of the slab, per sample
    __shared__ __attribute__((aligned(16))) double s_u[QG][512 + 4];       // u at offset 2 (Phi q before a row's slab)
    __shared__ __attribute__((aligned(16))) double s_g[QG][512];           // M u^n + dt F
    __shared__ __attribute__((aligned(16))) double s_fdt[QG][512];         // dt F
    __shared__ __attribute__((aligned(16))) double s_coef[QG][QRS][4];     // lo, di, up, R of the slab's rows
    __shared__ __attribute__((aligned(16))) double s_q[QG][QN];
    __shared__ double s_unext[QG];                                         // u of the first row of the NEXT slab
    __shared__ int s_act[QG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction: tile and row arithmetic on the scalar unit
    const int N = a.N, n = a.n, NPAD = a.NPAD;
    const double h = (a.x[N - 1] - a.x[0]) / (double)(N - 1);
    const int nslab = (N + QRS - 1) / QRS;
    double* Tw = s_T + w * QTJ;                       // this wave's sample
    double (*Sw)[QSYS] = reinterpret_cast<double (*)[QSYS]>(Tw);     // reduced system, parked over the dead tangent rows
    // lane roles: projection operand (k, blk, t), tangent result (i, blk, j)
    const int pk = lane >> 4, pblk = (lane >> 2) & 3, pt = lane & 3;

    for (int e = tid; e < QG * QTJ; e += 256) s_T[e] = 0.0;        // ring slots are read before their first write (row -1 of slab 0): keep them finite
    const int ngroups = (a.B + QG - 1) / QG;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int slot = grp * QG + w;
        const bool valid = slot < a.B;
        const int slotc = valid ? slot : a.B - 1;      // a padding wave computes on a copy of the last sample, stores nothing
        const int sb = a.order ? a.order[slotc] : slotc;
        const int smp = sb;
        const double mu1 = a.mu1[sb], mu2 = a.mu2[sb];
        double* hist = a.hist + (size_t)sb * (size_t)(a.nsteps + 1) * (size_t)N;
        __syncthreads();                               // the previous group is done with LDS
        // ---- per-sample constants (compute_forcing_vector :427-461) and the initial state, wave-local ------------
        for (int i = lane; i < 512; i += 64) {
            double frPrev = 0.0, fl = 0.0, u = 0.0;
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
                }
                u = a.u0[(size_t)sb * N + i];
                if (valid) hist[i] = u;
            }
            s_fdt[w][i] = a.dt * (frPrev + fl);
            s_u[w][i + 2] = u;
        }
        if (lane < 4) s_u[w][lane < 2 ? lane : 512 + lane] = 0.0;       // halos [0], [1], [514], [515]
        int flags = 0, info_out = 0;
        long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tick = kQT ? (long long)__builtin_amdgcn_s_memtime() : 0;
        auto lap = [&](int i) {
            if constexpr (kQT) {
                const long long now = (long long)__builtin_amdgcn_s_memtime();
                cyc[i] += now - tick;
                tick = now;
            }
        };
        int npass = 0;
        // T rows of a row group (4 mesh rows) for the four samples.  H3 of a mesh row is symmetric: only its 55 upper 4 x 4 blocks
        // (a <= b) are stored and streamed -- 3.7 MB instead of 6.5 MB per pass through the CU's vector-memory path, which bounded
        // this phase (first version: every (tile, k chunk) operand loaded, 48 B/clk/CU).  Block (a, b) feeds tile a as it stands
        // (k chunk b) and, if a < b, tile b transposed (k chunk a): the transposed operand wants the value of lane (i, blk, k) in
        // lane (k, blk, i), two ds_bpermute.  The blocks are consumed in storage order, two per 16-byte slot, against the ten
        // tile accumulators (two chains each: even / odd k chunk), so a slot is dead after at most four matrix instructions and
        // the ring of QW slots in flight runs seamlessly from one row group into the wave's next one.
        double2 hs[QW];
        double sd[QNB], sdn[QNB];                                             // accumulator seeds (Phi) of this group / the next one
        auto slot_ptr = [&](int g) {
            const int rg = g < 0 ? 0 : (g >= a.NG ? a.NG - 1 : g);            // clamped: the result of such a group is discarded
            return reinterpret_cast<const double2*>(a.H3f) + (size_t)rg * QP2 * 64 + lane;
        };
        auto load_seeds = [&](int g, double (&dst)[QNB]) {
            const int rg = g < 0 ? 0 : (g >= a.NG ? a.NG - 1 : g);
            const double* pp = a.Phif + (size_t)rg * QNB * 16 + (lane >> 2);
#pragma unroll
            for (int c = 0; c < QNB; ++c) dst[c] = pp[c * 16];
        };
        auto start_stream = [&](int g) {                                      // the first QW slots and the seeds of group g
            const double2* hp = slot_ptr(g);
#pragma unroll
            for (int p2 = 0; p2 < QW; ++p2) hs[p2] = hp[p2 * 64];
            load_seeds(g, sd);
        };
        auto tangent_group = [&](int g, int gnext, const double (&bq)[QKC]) {
            const int ti = lane >> 4, tblk = (lane >> 2) & 3, tj = lane & 3;
            const bool inside = g >= 0 && g < a.NG;
            const int trsrc = (16 * (lane & 3) + 4 * tblk + ti) << 2;         // byte index of the lane whose value this lane takes in a transpose
            const double2* hp = slot_ptr(g);
            const double2* hn = slot_ptr(gnext);
            load_seeds(gnext, sdn);
            double d[QNB][2];
            double pq = 0.0;                            // (Phi q)[row 4 g + blk][sample j]: the seeds ARE Phi, and bq[c] = q_j[4 c + i]
#pragma unroll
            for (int c = 0; c < QNB; ++c) { d[c][0] = sd[c]; d[c][1] = 0.0; pq = __builtin_fma(sd[c], bq[c], pq); }
            pq += from_lane_rot(pq, (lane ^ 16) << 2);                       // sum over i = lane bits 4, 5
            pq += from_lane_rot(pq, (lane ^ 32) << 2);
            if (ti == 0 && g >= 0 && 4 * g + tblk < 512) s_u[tj][2 + 4 * g + tblk] = inside ? pq : 0.0;      // seeds u = 1/2 (Phi q + T q) of the row's slab
            // transposes run one slot ahead of the products that consume them (a ds_bpermute result is ~100 clocks away:
            // computed in the slot's own turn each slot stalled for it, 2 k clocks per group)
            auto transposes = [&](int p2, double (&tr)[2]) {
                if (p2 < QP2) {
                    const double2 v2 = hs[p2 % QW];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int p = 2 * p2 + e;
                        if (p < QPAIRS && qrow(p) != qcol(p)) tr[e] = from_lane_rot(e ? v2.y : v2.x, trsrc);
                    }
                }
            };
            double trn[2] = {0.0, 0.0};
            transposes(0, trn);
#pragma unroll
            for (int p2 = 0; p2 < QP2; ++p2) {
                const double2 v2 = hs[p2 % QW];
                const double tr[2] = {trn[0], trn[1]};
                transposes(p2 + 1, trn);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = 2 * p2 + e;
                    if (p < QPAIRS) {
                        const int ra = qrow(p), cb = qcol(p);
                        d[ra][cb & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(e ? v2.y : v2.x, bq[cb], d[ra][cb & 1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = 2 * p2 + e;
                    if (p < QPAIRS && qrow(p) != qcol(p)) {
                        const int ra = qrow(p), cb = qcol(p);
                        d[cb][ra & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(tr[e], bq[ra], d[cb][ra & 1], 0, 0, 0);
                    }
                }
                // the slot is dead: refill it with slot p2 + QW of this group, or of the wave's next one (no branch: a join would drain the ring)
                hs[p2 % QW] = (p2 + QW < QP2) ? hp[(p2 + QW) * 64] : hn[(p2 + QW - QP2) * 64];
            }
            // result lane 16 i + 4 blk + j: T[sample j][row 4 g + blk][column 4 c + i]; groups outside the mesh are zero rows
            double* trow = s_T + tj * QTJ + (4 * ((g + 1) % QRING) + tblk) * QTS + ti;
#pragma unroll
            for (int c = 0; c < QNB; ++c) {
                trow[4 * c] = inside ? d[c][0] + d[c][1] : 0.0;
                sd[c] = sdn[c];
            }
        };
        // mesh row i (>= -4) -> row of the ring of tangent rows in LDS (group g sits in slot (g + 1) mod 18)
        auto ring_row = [](int i) { return 4 * (((i + 4) >> 2) % QRING) + ((i + 4) & 3); };

        for (int step = 0; step < a.nsteps; ++step) {
            // ---- g = M u^n + dt F (`M @ U[:, m] + At*F`, :1144), q = Phi^T u^n (:1129); wave-local ------------------
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double q = 0.0;                                // lane a < n holds q_a
            {
                double uu[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int i = lane + 64 * m;
                    const double um = s_u[w][i + 1], u0 = s_u[w][i + 2], ur = s_u[w][i + 3];
                    uu[m] = u0;
                    double g = 0.0;
                    if (i < N) {
                        if (a.nonuniform) {
                            double v = 0.0;
                            if (i > 0) v = (a.x[i] - a.x[i - 1]) / 6.0 * __builtin_fma(2.0, u0, um);
                            if (i < N - 1) v = __builtin_fma((a.x[i + 1] - a.x[i]) / 6.0, __builtin_fma(2.0, u0, ur), v);
                            g = v + s_fdt[w][i];
                        } else {
                            double acc;
                            if (i == 0) acc = __builtin_fma(2.0, u0, ur);
                            else if (i == N - 1) acc = __builtin_fma(2.0, u0, um);
                            else acc = __builtin_fma(4.0, u0, um) + ur;
                            g = __builtin_fma(h / 6.0, acc, s_fdt[w][i]);
                        }
                    }
                    s_g[w][i] = g;
                }
                for (int c0 = 0; c0 < QN; c0 += 4) {
                    double f[4][8];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const double* pc = a.PhiT + (size_t)(c0 + cc) * NPAD + lane;
#pragma unroll
                        for (int m = 0; m < 8; ++m) f[cc][m] = (64 * m < NPAD) ? pc[64 * m] : 0.0;
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        double p = 0.0;
#pragma unroll
                        for (int m = 0; m < 8; ++m) p = __builtin_fma(f[cc][m], uu[m], p);
                        p = wave_sum(p);
                        q = (lane == c0 + cc) ? p : q;
                    }
                }
            }
            int k = 0;
            bool act = valid && info_out == 0;             // a sample with a singular reduced system stays frozen
            lap(7);
            while (true) {
                ++npass;
                // ---- pass start: publish q and the activity flag (Phi q, the seed of the decode, comes out of the tangent stream) ----
                if (lane < QN) s_q[w][lane] = q;
                if (lane == 0) s_act[w] = act ? 1 : 0;
                start_stream(w == 0 ? 0 : 1 + 4 * w);
                lap(0);
                __syncthreads();
                lap(5);
                const bool any = (s_act[0] | s_act[1] | s_act[2] | s_act[3]) != 0;     // workgroup-uniform
                // B operand of the tangent: lane 16 k + 4 blk + j holds q_j[4 kc + k]
                double bq[QKC];
#pragma unroll
                for (int kc = 0; kc < QKC; ++kc) bq[kc] = s_q[pt][4 * kc + pk];
                double acc[NACC];
#pragma unroll
                for (int p = 0; p < NACC; ++p) acc[p] = 0.0;

                for (int slab = 0; slab < nslab; ++slab) {
                    const int r0 = slab * QRS;
                    // ---- tangent rows of the row groups 16 slab + 1 .. 16 slab + 16 for the four samples, four groups per wave ----------
                    // (the LDS ring still holds groups 16 slab - 1 and 16 slab from the previous slab: rows r0 - 1 .. r0 + 64 are
                    // what the decode and the projection of rows r0 .. r0 + 63 read)
                    {
                        const int g0 = 16 * slab + 1 + 4 * w;
                        const int first = (w == 0) ? 0 : 1 + 4 * w;                        // this wave's first group of a pass
                        if (slab == 0 && w == 0) tangent_group(0, g0, bq);
#pragma unroll 1
                        for (int m = 0; m < 4; ++m)
                            tangent_group(g0 + m, m < 3 ? g0 + m + 1 : (slab + 1 < nslab ? g0 + 16 : first), bq);
                    }
                    lap(1);
                    __syncthreads();
                    lap(2);
                    // ---- decode of this wave's sample, rows [r0, r0 + 64]:  u = 1/2 (Phi q + T q)  (:1116-1118) -------------
                    {
                        const double* trow = Tw + ring_row(r0 + lane) * QTS;
                        double s = 0.0;
#pragma unroll
                        for (int c4 = 0; c4 < QN / 4; ++c4) {
                            const double2 t0 = *reinterpret_cast<const double2*>(trow + 4 * c4);
                            const double2 t1 = *reinterpret_cast<const double2*>(trow + 4 * c4 + 2);
                            const double2 q0 = *reinterpret_cast<const double2*>(&s_q[w][4 * c4]);       // broadcast reads
                            const double2 q1 = *reinterpret_cast<const double2*>(&s_q[w][4 * c4 + 2]);
                            s = __builtin_fma(t0.x, q0.x, s);
                            s = __builtin_fma(t0.y, q0.y, s);
                            s = __builtin_fma(t1.x, q1.x, s);
                            s = __builtin_fma(t1.y, q1.y, s);
                        }
                        const int i = r0 + lane;
                        s_u[w][2 + i] = (i < N) ? 0.5 * (s_u[w][2 + i] + s) : 0.0;
                        // the row just beyond the slab (its own slab has not come yet: Phi q stays in s_u)
                        const int inx = r0 + QRS;
                        double v = (lane < QN) ? Tw[ring_row(r0 + QRS) * QTS + lane] * q : 0.0;
                        v = wave_sum(v);
                        if (lane == 0) s_unext[w] = (inx < N) ? 0.5 * (s_u[w][2 + inx] + v) : 0.0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    lap(3);
                    if (act) {
                        // ---- assembly: A(u), R(u) of row r0 + lane (no SUPG term in this variant, :1142) --------------------
                        {
                            const int i = r0 + lane;
                            const MeshConst mc = make_mesh_const(h, a.dt, a.E, 0);
                            const double um = s_u[w][i + 1], u0 = s_u[w][i + 2];
                            const double ur = (lane == QRS - 1) ? s_unext[w] : s_u[w][i + 3];
                            double lo, di, up, R;
                            rom_assemble_row(i, N, um, u0, (i + 1 < N) ? ur : 0.0, (i < N) ? s_g[w][i] : 0.0, 0.0, 0.0, mu1, mc,
                                             a.nonuniform, a.x, a.dt, a.E, lo, di, up, R);
                            *reinterpret_cast<double2*>(&s_coef[w][lane][0]) = make_double2(lo, di);
                            *reinterpret_cast<double2*>(&s_coef[w][lane][2]) = make_double2(up, R);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        // ---- projection: four steps of 16 rows; lane (k, blk, t): row 16 st + 4 k + blk, columns 10 t .. 10 t + 9 ----
#pragma unroll 1
                        for (int st = 0; st < QRS / 16; ++st) {
                            const int rl = 16 * st + 4 * pk + pblk;               // row within the slab
                            const double2 c01 = *reinterpret_cast<const double2*>(&s_coef[w][rl][0]);
                            const double2 c23 = *reinterpret_cast<const double2*>(&s_coef[w][rl][2]);
                            const double lo = c01.x, di = c01.y, up = c23.x, R = c23.y;
                            const double* trb = Tw + ring_row(r0 + rl - 1) * QTS + 10 * pt;
                            const double* trm = Tw + ring_row(r0 + rl) * QTS + 10 * pt;
                            const double* tra = Tw + ring_row(r0 + rl + 1) * QTS + 10 * pt;
                            double Tm[QNB], Y[QNB];
#pragma unroll
                            for (int c2 = 0; c2 < QNB / 2; ++c2) {
                                const double2 tb = *reinterpret_cast<const double2*>(trb + 2 * c2);
                                const double2 tm = *reinterpret_cast<const double2*>(trm + 2 * c2);
                                const double2 ta = *reinterpret_cast<const double2*>(tra + 2 * c2);
                                Tm[2 * c2] = tm.x; Tm[2 * c2 + 1] = tm.y;
                                Y[2 * c2] = __builtin_fma(up, ta.x, __builtin_fma(di, tm.x, lo * tb.x));
                                Y[2 * c2 + 1] = __builtin_fma(up, ta.y, __builtin_fma(di, tm.y, lo * tb.y));
                            }
                            const double X = (pt == 0) ? R : 0.0;                // extra B block [R, 0, 0, 0]
                            int p = 0;
                            if constexpr (GAL) {
#pragma unroll
                                for (int ca = 0; ca < QNB; ++ca) {
#pragma unroll
                                    for (int cb = 0; cb < QNB; ++cb, ++p)
                                        acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(Tm[ca], Y[cb], acc[p], 0, 0, 0);
                                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(Tm[ca], X, acc[p], 0, 0, 0);
                                    ++p;
                                }
                            } else {
#pragma unroll
                                for (int ca = 0; ca < QNB; ++ca) {
#pragma unroll
                                    for (int cb = ca; cb < QNB; ++cb, ++p)
                                        acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(Y[ca], Y[cb], acc[p], 0, 0, 0);
                                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(Y[ca], X, acc[p], 0, 0, 0);
                                    ++p;
                                }
                            }
                        }
                    }
                    lap(4);
                    __syncthreads();                       // every wave is done with this slab's tangent rows
                    lap(5);
                }
                if (!any) break;                           // that was the sweep for u(q_K): U[:, m+1] = u  (:1173)
                if (act) {
                    // ---- reduced system: sum the four block partials, park Ar | br (wave-private, over the dead tangent rows) ----
                    {
                        const int oi = lane >> 4, oj = lane & 3;
                        const bool writer = ((lane >> 2) & 3) == 3;
                        int p = 0;
#pragma unroll
                        for (int ca = 0; ca < QNB; ++ca) {
#pragma unroll
                            for (int cb = (GAL ? 0 : ca); cb <= QNB; ++cb, ++p) {
                                double v = acc[p];
                                v += dpp_mov<0x114>(v);          // row_shr:4
                                v += dpp_mov<0x118>(v);          // row_shr:8 -> lanes with blk == 3 hold the sum
                                if (cb < QNB) { if (writer) Sw[10 * oi + ca][10 * oj + cb] = v; }
                                else if (writer && oj == 0) Sw[10 * oi + ca][QN] = v;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // ---- solve(Ar, -br) (:1161): lane = row, Gauss-Jordan with partial pivoting ---------------------------------
                    // np.linalg.solve is LU with partial pivoting, and here it does leave the diagonal (far from the converged
                    // state the tangent columns are not near-orthonormal: the first version ran bg_rom_run's guarded pivot-free
                    // elimination and handed nearly every sample back).  Rows never move: the pivot of step kk is the not-yet-
                    // used lane with the largest |a_kk| (top 32 bits, as lu_pivoted_wave), its row is broadcast with v_readlane;
                    // every other row -- used or not -- is eliminated, so what is left is a permuted diagonal system.
                    double row[QN + 1];
                    {
                        // `lo` is opaque to the optimiser: left to itself it hoists the 80 per-lane LDS addresses, the 40 mirror
                        // masks and the 40 identity constants of this load out of every loop, spills them at kernel entry and
                        // reloads them here one scratch round trip at a time (40 k clocks per solve)
                        int lo = lane;
                        asm volatile("" : "+v"(lo));
                        const int r = lo < QN ? lo : 0;
                        if (lo >= n && lo < QN) Sw[lo][lo] = 1.0;               // identity padding (the padded rows and columns are exact zeros)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int rm = r % 10;
                        const double* direct = &Sw[r][0];
                        const double* mirror = &Sw[0][r];
#pragma unroll
                        for (int c = 0; c < QN; ++c) {
                            const double vd = direct[c], vm = GAL ? 0.0 : mirror[c * QSYS];
                            row[c] = (GAL || rm <= (c % 10)) ? vd : vm;         // LSPG: the lower blocks by symmetry
                        }
                        row[QN] = -direct[QN];
                    }
                    bool used = lane >= QN;                 // lanes beyond the system never pivot
                    int my_step = -1, sing = 0;
                    double diag = 1.0;
#pragma unroll
                    for (int kk = 0; kk < QN; ++kk) {
                        const unsigned key = used ? 0u : (((unsigned)__double2hiint(row[kk]) & 0x7fffffffu) + 1u);
                        const unsigned best = wave_max_u32(key);
                        const unsigned long long cand = __ballot(key == best && !used);
                        const int pl = __builtin_ctzll(cand);                   // lowest candidate lane
                        const double piv = readlane_f64(row[kk], pl);
                        sing = (piv == 0.0 && sing == 0) ? kk + 1 : sing;
                        const double rp = rcp(piv);
                        const bool is_p = lane == pl;
                        const double m = is_p ? 0.0 : row[kk] * rp;
                        diag = is_p ? piv : diag;
#pragma unroll
                        for (int c = kk + 1; c <= QN; ++c) row[c] = __builtin_fma(-m, readlane_f64(row[c], pl), row[c]);
                        used = used | is_p;
                        my_step = is_p ? kk : my_step;
                    }
                    // the lane that pivoted at step kk holds x_kk: put it into lane kk through this wave's (dead) coefficient rows
                    if (my_step >= 0) s_coef[w][my_step][0] = row[QN] * rcp(diag);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const double dq = (lane < n) ? s_coef[w][lane < QN ? lane : 0][0] : 0.0;
                    if (sing != 0) {                        // exactly singular: numpy raises LinAlgError at :1161
                        info_out = sing;
                        act = false;
                    } else {
                        // ---- q += dq, rel = |dq| / max(1e-14, |q|), stop when rel < tol (:1162-1169) ------------------------
                        q += dq;
                        double nd, nq;
                        wave_sum2(dq * dq, q * q, nd, nq);
                        const double rel = sqrt(nd) / fmax(1e-14, sqrt(nq));
                        ++k;
                        if (!(rel - rel == 0.0)) flags |= BG_FLAG_NONFINITE;
                        if (rel < a.tol) act = false;
                        else if (k >= a.max_it) { act = false; flags |= BG_FLAG_HIT_CAP; }   // "Newton did not converge" (:1171)
                    }
                }
                lap(6);
            }
            lap(7);
            // ---- U[:, m+1] = u (:1173): one coalesced row per sample ---------------------------------------------------------
            if (valid) {
                double* hrow = hist + (size_t)(step + 1) * N;
                for (int i = lane; i < N; i += 64) hrow[i] = s_u[w][i + 2];
                if (lane == 0) a.iters[(size_t)smp * a.nsteps + step] = k;
            }
        }
        if (valid && lane == 0) {
            a.flags[smp] = flags;
            a.info[smp] = info_out;
            if (kQT && a.nsteps >= 10) {                  // kilo-clocks per phase and the number of passes, in place of the counts
                for (int i = 0; i < 8; ++i) a.iters[(size_t)smp * a.nsteps + i] = (int)(cyc[i] >> 10);
                a.iters[(size_t)smp * a.nsteps + 8] = npass;
            }
        }
    }
}

}  // namespace

extern "C" {

// Largest n and N of bg_quad_rom_run.
int bg_quad_rom_max_n(void) { return QN; }

// Element counts of the two operand copies bg_quad_rom_run reads (the caller builds them once per basis).
long long bg_quad_rom_h3f_elems(int N) { return N < 2 ? 0 : (long long)((N + 3) / 4) * QP2 * 64 * 2; }
long long bg_quad_rom_phif_elems(int N) { return N < 2 ? 0 : (long long)((N + 3) / 4) * QNB * 16; }

int bg_quad_rom_run(int N, int B, int n, int nsteps, int projection, const double* x, const double* PhiT, const double* Phif,
                    const double* H3f, const double* u0, const double* mu1, const double* mu2, double dt, double E, double tol,
                    int max_it, int options, double* hist, int32_t* iters, int32_t* flags, int32_t* info, const int32_t* order,
                    void* stream)
{
    if (N < 2 || B < 0 || n < 1 || nsteps < 0 || max_it < 1 || !(dt > 0.0)) return BG_ERR_BAD_ARG;
    if (projection != BG_PROJ_GALERKIN && projection != BG_PROJ_LSPG) return BG_ERR_PROJECTION;
    if (N > 512) return BG_ERR_UNSUPPORTED_N;
    if (n > QN) return BG_ERR_UNSUPPORTED_R;
    if (B == 0) return BG_OK;
    if (!x || !PhiT || !Phif || !H3f || !u0 || !mu1 || !mu2 || !hist || !flags || !info || (nsteps > 0 && !iters)) return BG_ERR_BAD_ARG;
    QuadRunArgs a;
    a.x = x; a.PhiT = PhiT; a.Phif = Phif; a.H3f = H3f; a.u0 = u0; a.mu1 = mu1; a.mu2 = mu2; a.hist = hist; a.iters = iters;
    a.flags = flags; a.info = info; a.order = order; a.dt = dt; a.E = E; a.tol = tol; a.N = N; a.NPAD = ((N + 63) / 64) * 64; a.NG = (N + 3) / 4;
    a.B = B; a.n = n; a.nsteps = nsteps; a.max_it = max_it; a.nonuniform = (options & BG_OPT_NONUNIFORM) ? 1 : 0;
    const int cus = device_cu_count();
    const int groups = (B + QG - 1) / QG;
    const int grid = groups < cus ? groups : cus;
    hipStream_t st = (hipStream_t)stream;
    if (projection == BG_PROJ_GALERKIN)
        hipLaunchKernelGGL((quad_fused_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((quad_fused_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    return check_launch();
}

}  // extern "C"
