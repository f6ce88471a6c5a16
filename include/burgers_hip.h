/* burgers_hip.h -- C ABI of libburgers_hip.so (MI355X / gfx950).
 *
 * The reference (SADPR/1D-Burgers-Equation-ROMs) is pure Python on this path and
 * has no FFI of its own; each entry point below names the reference routine whose
 * inner loop it replaces (file:line relative to the reference checkout).  The
 * reference-side binding (a ctypes stub inside FEM/fem_burgers.py) is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) unless
 *     it is marked "host";
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream), allocate nothing, keep no global state and are thread-safe;
 *   - return value: BG_OK (0) or a negative BG_ERR_* code; nothing throws;
 *   - arrays are dense, row-major, float64 unless stated; "history" arrays are
 *     time-major per sample, hist[b][t][i] (the reference's (N, nT+1) C-order layout
 *     is produced by bg_transpose_batched);
 *   - one wavefront owns one (mu1, mu2) sample for the whole time loop.
 */
#ifndef BURGERS_HIP_H
#define BURGERS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_ABI_VERSION 1

enum {
    BG_OK = 0,
    BG_ERR_BAD_ARG = -1,        /* null pointer, negative size, ...                       */
    BG_ERR_UNSUPPORTED_N = -2,  /* N outside the range the wave-per-sample kernels cover  */
    BG_ERR_NONUNIFORM = -3,     /* mesh is not uniform (fast path only handles linspace)  */
    BG_ERR_LAUNCH = -4,         /* hipLaunchKernel failed; see bg_last_hip_error()        */
    BG_ERR_UNSUPPORTED_R = -5,  /* reduced dimension outside the supported range          */
    BG_ERR_PROJECTION = -6,     /* unknown projection enum                                */
    BG_ERR_WORKSPACE = -7       /* caller-provided workspace too small                    */
};

enum { BG_PROJ_GALERKIN = 0, BG_PROJ_LSPG = 1 };

/* option bits of the `supg` / `options` argument of the assembly-carrying entry points */
enum { BG_OPT_SUPG = 1,        /* include the SUPG vector (fom_burgers, pod_prom_burgers, pod_ann_prom) */
       BG_OPT_NONUNIFORM = 2,  /* x is not a linspace: use the per-element-length kernels          */
       BG_OPT_W_COLMAJOR = 4,  /* bg_rom_reduce*: W is [r][N] (per sample), not [N][r]             */
       BG_OPT_MFMA_16X16 = 8,  /* bg_rom_reduce*: use the v_mfma_f64_16x16x4 kernel for every r (A/B timing, tests) */
       BG_OPT_FORCE_PIVOTED = 16, /* bg_rom_run: take the partial-pivoting branch of the reduced solve every time (tests) */
       BG_OPT_NO_TANGENT_REUSE = 32, /* bg_ann_rom_run: evaluate the closure at every step start even when its float32 input is unchanged (tests) */
       BG_OPT_FOM_WIDE = 64,   /* bg_fom_run: one WORKGROUP per sample also for 64 < N <= 1536 (uniform mesh); measured slower than the default there */
       BG_OPT_FOM_WAVE = 128   /* bg_fom_run: one WAVEFRONT per sample (the default for N <= 1536; overrides BG_OPT_FOM_WIDE) */ };

/* bg_rom_run: transient value of info[b] between its two kernels (never seen by the caller);
 * bg_rom_run_wide: info[b] of a sample the caller must redo with a pivoting solve */
#define BG_INFO_NEEDS_PIVOTING (-1)

/* per-sample status bits written to `flags` */
enum { BG_FLAG_HIT_CAP = 1, BG_FLAG_NONFINITE = 2 };

#define BG_COUNTER_SLOTS 16   /* partial counters of bg_lu_solve_update ...          */
#define BG_COUNTER_STRIDE 32  /* ... one per 128-byte line (stride in int32 elements) */

/* activation kinds of bg_mlp_act_jvp */
enum { BG_ACT_NONE = 0, BG_ACT_ELU = 1, BG_ACT_RELU = 2, BG_ACT_TANH = 3 };

/* kernels of bg_rbf_eval */
enum { BG_RBF_GAUSSIAN = 0, BG_RBF_IMQ = 1 };

int bg_abi_version(void);
const char *bg_strerror(int code);
/* hipError_t of the most recent failed launch on the calling thread (0 if none). */
int bg_last_hip_error(void);

/* Largest N of bg_fom_run / bg_fom_assemble / bg_tridiag_solve: 8192.  N <= 1536 runs one wavefront per
 * sample (rows per lane <= 24), 1536 < N <= 8192 one 256-thread workgroup per sample (rows per thread <= 32);
 * at 24 rows per lane and 32 per thread part of the state spills to AGPRs / scratch.  bg_fd_run switches to
 * the workgroup form above N = 2048 (same limit, 8192). */
int bg_fom_max_n(void);

/* ---------------------------------------------------------------------------------
 * bg_fom_run -- batched replacement of FEMBurgers.fom_burgers
 *   reference: FEM/fem_burgers.py:646-707 (time loop + Picard loop), with
 *   compute_convection_matrix :389-425, compute_forcing_vector :427-461,
 *   compute_supg_term :500-581 and scipy spsolve :692 fused into one kernel.
 *
 *   x      [N]               mesh nodes, strictly increasing; pass BG_OPT_NONUNIFORM in `supg` when
 *                            they are not a linspace (the host knows: X is host data in the reference)
 *   u0     [B][N]            initial state per sample
 *   mu1,mu2[B]               Dirichlet value u(0,t) and source exponent per sample
 *   hist   [B][nsteps+1][N]  hist[b][0] = u0[b]; hist[b][t+1] = state after step t
 *   iters  [B][nsteps]       Picard iterations taken per step (the reference's k)
 *   flags  [B]               BG_FLAG_* bits
 *   tol, max_it              the reference hard-codes 1e-6 and 20 (:663)
 *   supg                     option bits: BG_OPT_SUPG (fom_burgers has it) | BG_OPT_NONUNIFORM
 * --------------------------------------------------------------------------------- */
int bg_fom_run(int N, int B, int nsteps, const double *x, const double *u0, const double *mu1,
               const double *mu2, double dt, double E, double tol, int max_it, int supg,
               double *hist, int32_t *iters, int32_t *flags, void *stream);

/* bg_fom_run_traced -- bg_fom_run that also stores the error of every Picard iteration, the value the reference prints
 *   per iteration (`print(f"Iteration: {k}, Error: {error_U}")`, FEM/fem_burgers.py:664; error_U = ||dU|| / ||U1|| of :698).
 *   errs [B][nsteps][max_it]: errs[b][t][k] = error after iteration k of step t (entries k >= iters[b][t] untouched).
 *   Same results as bg_fom_run; a separate kernel instantiation, N <= 1536 (BG_ERR_UNSUPPORTED_N beyond). */
int bg_fom_run_traced(int N, int B, int nsteps, const double *x, const double *u0, const double *mu1,
                      const double *mu2, double dt, double E, double tol, int max_it, int supg,
                      double *hist, int32_t *iters, int32_t *flags, double *errs, void *stream);

/* ---------------------------------------------------------------------------------
 * bg_fom_assemble -- one Picard assembly, for inspection and tests
 *   reference: FEM/fem_burgers.py:666-689 (C, S, F, A with Dirichlet row, b, R).
 *   uk, un [B][N]; outputs lo/di/up/rhs [B][N]: the three diagonals of A(u_k)
 *   (lo[.][0] = up[.][N-1] = 0) and rhs = b - A u_k = -R.
 * --------------------------------------------------------------------------------- */
int bg_fom_assemble(int N, int B, const double *x, const double *uk, const double *un,
                    const double *mu1, const double *mu2, double dt, double E, int supg,
                    double *lo, double *di, double *up, double *rhs, void *stream);

/* ---------------------------------------------------------------------------------
 * bg_tridiag_solve -- batched pivot-free tridiagonal solve, one wavefront per system
 *   reference: scipy.sparse.linalg.spsolve call at FEM/fem_burgers.py:692.
 *   lo/di/up/rhs [B][N] as produced by bg_fom_assemble; sol [B][N].
 * --------------------------------------------------------------------------------- */
int bg_tridiag_solve(int N, int B, const double *lo, const double *di, const double *up,
                     const double *rhs, double *sol, void *stream);

/* ---------------------------------------------------------------------------------
 * bg_transpose_batched -- out[b][c][r] = in[b][r][c]
 *   turns the time-major history [B][nT+1][N] into the reference's (N, nT+1) C-order
 *   snapshot layout per sample (FEM/fem_burgers.py:650, np.save at
 *   FEM/paper_training_stage.py:52-53).
 * --------------------------------------------------------------------------------- */
int bg_transpose_batched(int B, int rows, int cols, const double *in, double *out, void *stream);

/* =================================================================================
 * Projection ROMs: one batched Picard / Gauss-Newton iteration is
 *     bg_rom_reduce  ->  bg_lu_solve  ->  (family-specific update, host side GEMMs)
 * ================================================================================= */

/* Largest N / r the register-resident MFMA reduce kernel covers (N <= 512, r <= 47). */
int bg_rom_max_n(void);
int bg_rom_max_r(void);

/* bg_forcing_setup -- per-sample load constants, once per run
 *   reference: compute_forcing_vector FEM/fem_burgers.py:427-461 (the reference recomputes
 *   it every iteration, :673) and the f_gp terms of compute_supg_term :556-558.
 *   fdt[b][i] = dt * F_i(mu2_b);  hfs[b][e] = h_e * (f(gp1) + f(gp2)), hfs[b][N-1] = 0. */
int bg_forcing_setup(int N, int B, const double *x, const double *mu2, double dt, int options,
                     double *fdt, double *hfs, void *stream);

/* bg_mass_rhs -- g[b] = M u^n[b] + dt F[b], once per time step
 *   reference: `M @ U[:, n] + At*F` at FEM/fem_burgers.py:683 / :746 / :1141 / :1214. */
int bg_mass_rhs(int N, int B, const double *x, const double *un, const double *fdt, int options,
                double *g, void *stream);

/* bg_rom_reduce -- fused assembly + projection of one iteration, fp64 MFMA
 *   reference: FEM/fem_burgers.py:730-762 (pod_prom_burgers), :1134-1156
 *   (pod_quadratic_manifold, supg = 0), :1203-1233 (pod_ann_prom).
 *   W        basis or tangent, [N][r] row-major shared by all samples (w_stride = 0) or
 *            [B][N][r] with w_stride = N*r elements; with BG_OPT_W_COLMAJOR in `supg` each
 *            sample's block is [r][N] instead (what one GEMM dN^T [B*r x m] . U_s^T [m x N] yields)
 *   U        [B][N] current iterate u_k;  G from bg_mass_rhs;  hfs from bg_forcing_setup
 *   active   [B] int32 or NULL: samples with 0 are skipped and their outputs left untouched
 *   Ar       [B][r][r]: W^T A W (BG_PROJ_GALERKIN) or (A W)^T (A W) (BG_PROJ_LSPG)
 *   br       [B][r]:    W^T R   or (A W)^T R,  R = A u_k - b
 *   wtu      [B][r] or NULL: W^T u_k (the `Phi.T @ U0` of :770)
 *   Two kernels: v_mfma_f64_4x4x4_4b (r <= 40) and v_mfma_f64_16x16x4 (r <= 47).  BG_OPT_MFMA_16X16 in `supg`
 *   selects the second one for every r, for A/B timing and tests (no environment is read). */
int bg_rom_reduce(int N, int B, int r, int projection, const double *x, const double *W,
                  long long w_stride, const double *U, const double *G, const double *hfs,
                  const double *mu1, double dt, double E, int supg, const int32_t *active,
                  double *Ar, double *br, double *wtu, void *stream);

/* bg_rom_reduce_indexed -- bg_rom_reduce with a small set of bases shared among the samples: sample b
 * uses the block W + w_index[b] * w_stride (local POD: the basis of its cluster; reference
 * local_prom_burgers, FEM/fem_burgers.py:1011-1013).  A workgroup keeps the fragments in registers
 * while consecutive samples of its stride use the same block. */
int bg_rom_reduce_indexed(int N, int B, int r, int projection, const double *x, const double *W,
                          long long w_stride, const int32_t *w_index, const double *U, const double *G,
                          const double *hfs, const double *mu1, double dt, double E, int supg,
                          const int32_t *active, double *Ar, double *br, double *wtu, void *stream);

/* bg_lu_solve -- x[b] = solve(A[b], sign * rhs[b]), partial pivoting, n <= 64
 *   reference: np.linalg.solve(Ar, -br) at FEM/fem_burgers.py:767 / :1161 / :1237
 *   (LAPACK gesv).  info[b] = 0, or k+1 when the pivot of step k is exactly zero
 *   (numpy raises LinAlgError("Singular matrix") there). */
int bg_lu_solve(int n, int B, const double *A, const double *rhs, double sign, const int32_t *active,
                double *x, int32_t *info, void *stream);

/* bg_rom_reduce_lifted -- bg_rom_reduce for a shared basis Phi with the lift fused in:
 *   u_k = Phi q is formed from the register-resident basis, stored to U[b] and used for the
 *   assembly, so the state never round-trips through a GEMM.  reference: `U1 = Phi @ q` :773. */
int bg_rom_reduce_lifted(int N, int B, int r, int projection, const double *x, const double *Phi,
                         const double *q, double *U, const double *G, const double *hfs,
                         const double *mu1, double dt, double E, int supg, const int32_t *active,
                         double *Ar, double *br, double *wtu, void *stream);

/* bg_rom_lift -- U[b] = Phi q[b] only (end of a time step). */
int bg_rom_lift(int N, int B, int r, const double *x, const double *Phi, const double *q,
                const int32_t *active, double *U, void *stream);

/* bg_lu_solve_update -- bg_lu_solve(A, -rhs) fused with the reduced-coordinate update and the
 * convergence bookkeeping of one batched iteration.  Samples with active[b] == 0 are skipped.
 *   mode 1 (pod_prom_burgers :770-776):       q = wtu + dq;  err = |dq|/|q|;            go on while err > tol and k < max_it
 *   mode 2 (pod_quadratic_manifold :1161-69): q += dq;       err = |dq|/max(1e-14,|q|); stop when err < tol (cap max_it)
 *   mode 3 (pod_ann_prom :1237-1244):         q += dq;       err = |dq|/(|q|+1e-14);    go on while err > tol and k < max_it
 *   iters[b] += 1; active[b] <- go on; flags[b] |= BG_FLAG_*.
 *   counter [2][BG_COUNTER_SLOTS][BG_COUNTER_STRIDE] int32, element [.][s][0] used: row 0 sums to the
 *   number of samples still active, row 1 to the number of singular systems (sample b adds to slot
 *   b % SLOTS, one cache line per slot: 4096 atomics on one line cost 8 us); the caller zeroes it
 *   before the call and sums the slots. */
int bg_lu_solve_update(int n, int B, const double *A, const double *rhs, int mode, const double *wtu,
                       double *q, double *dq, double tol, int max_it, int32_t *active, int32_t *iters,
                       int32_t *flags, int32_t *counter, int32_t *info, void *stream);

/* ---------------------------------------------------------------------------------
 * bg_rom_run -- batched replacement of FEMBurgers.pod_prom_burgers, the WHOLE time loop on the device
 *   reference: FEM/fem_burgers.py:709-785.  One workgroup owns one sample for all time steps and Picard iterations:
 *   assembly, fp64-MFMA projection (as bg_rom_reduce), the r x r solve, q = Phi^T u + dq, the stopping test
 *   err = |dq|/|q| > tol and k < max_it (:729, :776) and the lift u = Phi q run with no kernel boundary and no host
 *   in between; HBM sees u0 once and one N-row history write per time step.
 *   Phi    [N][r] row-major (the reference's U_modes .npy layout), shared by all samples; N <= 512, r <= 40
 *          (bg_rom_run_max_r) -- beyond that use bg_rom_reduce + bg_lu_solve_update (BG_ERR_UNSUPPORTED_R / _N)
 *   u0, mu1, mu2, hist, iters, flags: as bg_fom_run (hist[b][0] = u0[b]; flags BG_FLAG_*)
 *   info   [B] (required): 0, or k+1 when the reduced matrix of a sample is exactly singular at elimination step k
 *          (np.linalg.solve raises LinAlgError there); that sample stops and its remaining history is undefined
 *   options BG_OPT_SUPG (pod_prom_burgers has it) | BG_OPT_NONUNIFORM | BG_OPT_FORCE_PIVOTED
 *   The reduced solve is np.linalg.solve's partial-pivoting elimination: as long as every multiplier stays <= 1 in
 *   modulus the pivot is the diagonal and no search is made (first kernel).  A sample in which a multiplier exceeds 1
 *   is marked in info and redone from u0 by a second kernel of the same call with the pivot search of bg_lu_solve;
 *   that kernel returns at once when nothing is marked.  BG_OPT_FORCE_PIVOTED sends every sample through it.
 *   order  NULL, or [B] int32 on the device: a permutation of 0 .. B-1 -- slot i of the launch works on sample order[i]
 *          (a scheduling hint, results are the same bit for bit: the samples are independent).  The workgroups are
 *          persistent, workgroup k of G takes the slots k, k + G, ... (G = min(B, 2 CUs); bg_rom_run_wide and
 *          bg_ann_rom_run alike with G = min(B, CUs) / min(B, 2 CUs)); in bg_quad_rom_run the slots 4 g .. 4 g + 3 share
 *          a workgroup AND its passes (every pass lasts until the slowest of the four has converged).  Iteration counts
 *          follow mu1 (correlation 0.99 on the bench sweep), so a caller that sorts the samples by mu1 and deals them out
 *          evenly (burgers_hip/rom.py::sample_order) removes the imbalance; measured at the bench sizes
 *          (tools/time_balance.py): quadratic + 2.5 %, POD-ANN + 1.8 %, r = 96 + 2.9 %, r = 40 + 0.5 %.  The reference-side
 *          binding passes NULL.
 * --------------------------------------------------------------------------------- */
int bg_rom_run_max_r(void);
int bg_rom_run(int N, int B, int r, int nsteps, int projection, const double *x, const double *Phi,
               const double *u0, const double *mu1, const double *mu2, double dt, double E, double tol,
               int max_it, int options, double *hist, int32_t *iters, int32_t *flags, int32_t *info,
               const int32_t *order, void *stream);

/* bg_rom_run_wide -- bg_rom_run for the thesis' larger bases, 40 < r <= 96 (bg_rom_run_wide_max_r), N <= 512
 *   reference: FEM/fem_burgers.py:709-785 with POD/modes/U_modes_tol_1e-04.npy (r = 96), POD/Results_thesis/prom_pod.py:35-58.
 *   Same arguments, outputs and semantics as bg_rom_run except:
 *   PhiP   [NPAD + 2][96], NPAD = N rounded up to 64 (bg_rom_run_wide_phi_elems(N) doubles, 16-byte aligned): Phi row i at
 *          row index i + 1, zero rows around and beyond the mesh, zero columns beyond r -- built once per basis by the caller;
 *   info   0, k + 1 for an exactly singular reduced system, or BG_INFO_NEEDS_PIVOTING for a sample whose pivot-free
 *          elimination met a multiplier above 1: the caller redoes that sample with a pivoting solve (there is no second
 *          kernel here; burgers_hip/rom.py sends it through the library path).
 *   The basis streams through LDS 64 mesh rows at a time, the accumulators of the reduced system are dealt to the four
 *   waves (csrc/rom_wide.hip).  options: BG_OPT_SUPG | BG_OPT_NONUNIFORM | BG_OPT_FORCE_PIVOTED (tests: every sample is
 *   handed back as if its elimination had needed a row exchange). */
int bg_rom_run_wide_max_r(void);
long long bg_rom_run_wide_phi_elems(int N);
int bg_rom_run_wide(int N, int B, int r, int nsteps, int projection, const double *x, const double *PhiP,
                    const double *u0, const double *mu1, const double *mu2, double dt, double E, double tol,
                    int max_it, int options, double *hist, int32_t *iters, int32_t *flags, int32_t *info,
                    const int32_t *order, void *stream);

/* =================================================================================
 * bg_fd_run -- batched replacement of FDBurgers.fom_burgers_newton (analytical Jacobian)
 *   reference: FD/fd_burgers.py:59-107 (time + Newton loops), residual :28-35, Jacobian :37-44,
 *   boundary values :19-22 (U[0] = mu1, U[-1] = U[-2]).  Central differences with the lagged
 *   artificial viscosity nu = 0.25 dx max|U|; stop on max|R| < tol or max|dU|/max|U| < tol.
 *   Arrays as in bg_fom_run; x must be the linspace(a, b, N) of the reference; N <= 8192.
 *   iters[b][t] = Newton solves taken in step t; flags: BG_FLAG_HIT_CAP when max_it ran out.
 * ================================================================================= */
int bg_fd_run(int N, int B, int nsteps, const double *x, const double *u0, const double *mu1,
              const double *mu2, double dt, double tol, int max_it, double *hist, int32_t *iters,
              int32_t *flags, void *stream);

/* ---------------------------------------------------------------------------------
 * Quadratic-manifold tangent, fused with the layout the MFMA reduce kernel wants
 *   reference: tangent(q) = Phi + H @ get_dQ_dq(q)   FEM/fem_burgers.py:1120-1123, :292-312
 *   H3 [N][n][NP] = H[i][pair(a,c)] * (1 + delta_ac), q [B][NP]: last dimension zero-padded to
 *                   NP = bg_rom_frag_pad(n) (a multiple of 8; built once / padded on the host side)
 *   bg_rom_frag_elems(N, r): doubles per sample of the fragment-major layout (0 if N > 512 or r > 40)
 *   bg_quad_tangent:        Wfrag[b] = Phi + H3 . q[b]   for every active sample
 *   bg_rom_reduce_frag:     bg_rom_reduce with W given in that layout (stride = bg_rom_frag_elems)
 * --------------------------------------------------------------------------------- */
/* bg_quad_features -- feat[b] = [q[b] | Q(q[b])], Q = the n(n+1)/2 unique products q_i q_j (j >= i) in the order of
 *   get_sym (FEM/fem_burgers.py:263-273): the left operand of the decode u = Phi q + H Q(q) (:1116-1118) as one matrix, so
 *   that the decode is ONE GEMM against [Phi^T; H^T].  q [B][n]; pair_i, pair_j [n(n+1)/2] int32 (row-major upper
 *   triangle); feat [B][n + n(n+1)/2]. */
int bg_quad_features(int B, int n, const double *q, const int32_t *pair_i, const int32_t *pair_j, double *feat,
                     void *stream);
long long bg_rom_frag_elems(int N, int r);
int bg_rom_frag_pad(int r);
int bg_quad_tangent(int N, int B, int n, const double *Phi, const double *H3, const double *q,
                    const int32_t *active, double *Wfrag, void *stream);
int bg_rom_reduce_frag(int N, int B, int r, int projection, const double *x, const double *Wfrag,
                       const double *U, const double *G, const double *hfs, const double *mu1, double dt,
                       double E, int supg, const int32_t *active, double *Ar, double *br, double *wtu,
                       void *stream);

/* bg_mlp_act_jvp -- activation stage of a forward-mode MLP evaluation, float32, in place
 *   reference: compute_ann_jacobian FEM/fem_burgers.py:1254-1275 (per-sample autograd Jacobian of the
 *   POD-ANN closure) and the model call at :1241; the linear layers are library GEMMs over B*n1 rows.
 *   z      [B][n1][h]: row 0 = W x (no bias yet), rows 1..n1-1 = W (dx/dq_k)
 *   bias   [h] or NULL
 *   out:   row 0 <- act(row 0 + bias); rows k >= 1 <- act'(row 0 + bias) * row k
 *   act    BG_ACT_NONE | BG_ACT_ELU (alpha) | BG_ACT_RELU | BG_ACT_TANH */
int bg_mlp_act_jvp(int B, int n1, int h, float *z, const float *bias, int act, float alpha, void *stream);

/* ---------------------------------------------------------------------------------
 * bg_quad_rom_run -- batched replacement of FEMBurgers.pod_quadratic_manifold, the WHOLE time loop on the device
 *   reference: FEM/fem_burgers.py:1081-1175 (decoder :1116-1118, tangent :1120-1123 with get_dQ_dq :292-312 folded
 *   in, Newton loop :1126-1173); called by Quadratic_manifold/quadratic_prom_simulation.py:49-55.
 *   One 256-thread workgroup owns FOUR samples for all time steps and iterations: tangent T = Phi + H3 q of the four
 *   samples on v_mfma_f64_4x4x4_4b (H3 streamed once per four sample-iterations), decode u = 1/2 (Phi q + T q) from
 *   the same rows, assembly, projection and the n x n solve per wave (csrc/quad_fused.hip).  No SUPG term (:1142).
 *   N <= 512, n <= bg_quad_rom_max_n() (40).  Operand copies, built once per basis by the caller (zero padded):
 *     PhiT [40][NPAD]               Phi^T, NPAD = N rounded up to 64
 *     Phif [NG][10][16]             Phi[4 rg + blk][4 c + i] at [rg][c][4 i + blk], NG = ceil(N / 4)
 *     H3f  [NG][28][64][2]          H3[i][a][c] = H[i][pair(a, c)] (1 + delta_ac) is symmetric in (a, c): only its upper 4 x 4
 *                                   blocks (A <= B, row-major, 55 per mesh-row group) are stored, two per 16-byte slot:
 *                                   block p = 2 slot + e of row group rg holds, at lane 16 k + 4 blk + i,
 *                                   H3[4 rg + blk][4 A + i][4 B + k]   (the A operand of the matrix instruction; block 55 = 0)
 *     sizes: bg_quad_rom_phif_elems(N), bg_quad_rom_h3f_elems(N) doubles.
 *   Outputs as bg_rom_run: hist [B][nsteps+1][N], iters [B][nsteps] (Newton iterations per step), flags [B]
 *   (BG_FLAG_HIT_CAP = "Newton did not converge" :1171, BG_FLAG_NONFINITE), info [B]: 0, or k + 1 when the reduced
 *   system met an exactly zero pivot at elimination step k (np.linalg.solve :1161 raises LinAlgError: so does the facade).
 *   options: BG_OPT_NONUNIFORM.
 * --------------------------------------------------------------------------------- */
int bg_quad_rom_max_n(void);
long long bg_quad_rom_h3f_elems(int N);
long long bg_quad_rom_phif_elems(int N);
int bg_quad_rom_run(int N, int B, int n, int nsteps, int projection, const double *x, const double *PhiT,
                    const double *Phif, const double *H3f, const double *u0, const double *mu1, const double *mu2,
                    double dt, double E, double tol, int max_it, int options, double *hist, int32_t *iters,
                    int32_t *flags, int32_t *info, const int32_t *order, void *stream);

/* ---------------------------------------------------------------------------------
 * bg_ann_rom_run -- batched replacement of FEMBurgers.pod_ann_prom, the WHOLE time loop on the device
 *   reference: FEM/fem_burgers.py:1177-1251 (loop), compute_ann_jacobian :1254-1275, model POD-ANN/pod_ann.py:38-56.
 *   One workgroup owns one sample for all time steps and Gauss-Newton iterations: assembly, fp64-MFMA projection of the
 *   tangent W = U_p + U_s dN (:1224), the n x n solve (:1237), q_p += dq with err = |dq| / (|q_p| + 1e-14) (:1238-1244),
 *   the closure N(q_p) with its input-Jacobian in one float32 forward-mode pass (the reference evaluates the model in
 *   float32 too), and the decode u = U_p q_p + U_s N(q_p) (:1242).  At the start of a time step q_p = U_p^T u (:1197).
 *   UT     [m8][N] row-major, m8 = (n + nbar) rounded up to a multiple of 8: rows 0 .. n-1 = U_p^T (the columns of U_p,
 *          contiguous), rows n .. n+nbar-1 = U_s^T, the rest zero; n <= 8, nbar <= 128, N <= 512.  One array because the
 *          decode and the tangent are ONE sweep over all n + nbar modes, eight at a time.
 *   the closure, layer l = 0 .. n_layers-1 (n_layers <= 8), all four arrays HOST arrays of length n_layers (widths:
 *   n_layers + 1, widths[0] = n, widths[n_layers] = nbar, every width <= 256):
 *     wt[l]    DEVICE pointer, 16-byte aligned, float32 [in4][ld] row-major = the TRANSPOSE of torch's Linear.weight
 *              zero-padded to in4 = widths[l] rounded up to 4 rows and ld = widths[l+1] rounded up to 8 columns (a thread
 *              fetches the weights of 8 outputs of one input as two 16-byte loads, unguarded)
 *     bias[l]  DEVICE pointer, float32 [widths[l+1]], or NULL
 *     acts[l]  BG_ACT_* applied after layer l, alphas[l] its ELU alpha
 *   limits are reported by bg_ann_rom_limits; a model outside them returns BG_ERR_UNSUPPORTED_R (use the per-iteration
 *   entry points bg_mlp_act_jvp + bg_rom_reduce + bg_lu_solve_update instead).
 *   u0, mu1, mu2, hist, iters, flags, info: as bg_rom_run; options BG_OPT_SUPG (pod_ann_prom has it) | BG_OPT_NONUNIFORM |
 *   BG_OPT_NO_TANGENT_REUSE.  The first-pass tangent of a time step is dN at float32(U_p^T u^n) (:1197, :1219); when those
 *   floats are bitwise the input of the previous step's last evaluation (the rule: u^n is that step's decode), the tangent is
 *   still on chip and the evaluation, which would reproduce it bit for bit, is skipped.
 *   The n x n solve is np.linalg.solve's elimination with the pivot search, always (one kernel): the columns of
 *   U_p + U_s dN are far from orthonormal and LAPACK does leave the diagonal on these systems.
 * --------------------------------------------------------------------------------- */
int bg_ann_rom_limits(int *max_n, int *max_nbar, int *max_width, int *max_layers);
int bg_ann_rom_run(int N, int B, int n, int nbar, int nsteps, int projection, const double *x, const double *UT,
                   const double *u0, const double *mu1, const double *mu2, int n_layers,
                   const int *widths, const float *const *wt, const float *const *bias, const int *acts,
                   const float *alphas, double dt, double E, double tol, int max_it, int options, double *hist,
                   int32_t *iters, int32_t *flags, int32_t *info, const int32_t *order, void *stream);

/* bg_decode_modes_bf16 -- the contraction of the non-intrusive POD-ANN decoder (bf16 tier of BASELINE config 5)
 *   reference: `Uhat = U_modes @ Qhat.T`, Non-Instrusive/predict_pod_ann.py:73-80, for a batch of (mu1, mu2) samples
 *   out[b][i][t] = sum_k Um[i][k] * Q[b * Nt + t][k]: bf16 operands, float32 accumulate, each result written once as
 *   float64 in the (N, Nt) C-order snapshot layout per sample.
 *   Um  [N][n] bf16 (raw 16-bit patterns), Q [B * Nt][n] bf16 (the MLP output), both 16-byte aligned; out [B][N][Nt]
 *   N a multiple of 32, n a multiple of 16, n <= 256 (BG_ERR_UNSUPPORTED_N / _R otherwise: use a library GEMM). */
int bg_decode_modes_bf16(int N, int n, int B, int Nt, const uint16_t *Um, const uint16_t *Q, double *out, void *stream);

/* bg_decode_mlp_bf16 -- the whole non-intrusive decoder in ONE kernel: MLP + contraction (bf16 tier of BASELINE config 5)
 *   reference: Non-Instrusive/predict_pod_ann.py:60-80 (standardised (mu1, mu2, t) -> model -> Qhat -> U_modes @ Qhat.T)
 *   A workgroup evaluates the MLP for its own 128 (sample, time level) columns -- activations as bf16 in LDS, bf16 MFMA with
 *   float32 accumulate, the rounding points of the PyTorch bf16 module (Linear output -> bf16, activation in float32 -> bf16)
 *   -- and contracts them with U_modes as bg_decode_modes_bf16 does: no activation or coefficient ever crosses HBM.
 *     z1 [B], z2 [B]  standardised mu1, mu2 (float64; rounded to bf16 through float32 in the kernel, as `.to(bfloat16)`)
 *     ztau [Nt]       standardised time levels
 *     layer l (host arrays of n_layers entries; DEVICE pointers inside):
 *       W[l]     [wout[l]][win[l]] bf16 row-major = torch Linear.weight zero padded: win[0] = 16 (the 3 inputs padded),
 *                win[l] = wout[l-1], every wout a multiple of 32 (padding features: zero rows, zero bias), wout[last] = n
 *       bias[l]  [wout[l]] bf16 or NULL;  acts[l] BG_ACT_* after layer l, alphas[l] its ELU alpha
 *   Um [N][n] bf16, out [B][N][Nt] float64.  N a multiple of 32, n a multiple of 32, widths <= 256, n_layers <= 8
 *   (BG_ERR_UNSUPPORTED_N / _R otherwise: run the model in PyTorch and call bg_decode_modes_bf16). */
int bg_decode_mlp_bf16(int N, int n, int B, int Nt, const uint16_t *Um, const double *z1, const double *z2, const double *ztau,
                       int n_layers, const int *win, const int *wout, const uint16_t *const *W, const uint16_t *const *bias,
                       const int *acts, const float *alphas, double *out, void *stream);

/* bg_jacobi_sweep -- n_steps steps of a one-sided (Hestenes) Jacobi SVD sweep, the accurate small core of
 * the snapshot SVD (reference: np.linalg.svd at POD/pod.py:84, build_quadratic_manifold.py:29).
 *   G      [m][ld] row-major: the m rows are orthogonalised in place by plane rotations
 *   J      [m][ld]: receives the same rotations (start from the identity)
 *   pairs  [n_steps][n_pairs][2] int32: disjoint row pairs of every step (round-robin ordering);
 *          an entry < 0 marks a bye
 *   tol    rows p, q are rotated when |g_p . g_q| > tol |g_p| |g_q|
 *   rotations [1] int32: += number of rotations applied (0 after a full sweep = converged)
 * After convergence the row norms of G are the singular values of the input, G[j]/|G[j]| its left
 * singular vectors (as rows), J its right singular vectors (as rows). */
int bg_jacobi_sweep(int m, int ld, double *G, double *J, const int32_t *pairs, int n_steps, int n_pairs,
                    double tol, int32_t *rotations, void *stream);

/* bg_rbf_eval -- kernel values and gradient factors of the scaled RBF closure of pod_rbf_prom
 *   reference: FEM/fem_burgers.py:160-260 (scaled gaussian / inverse-multiquadric closure and its Jacobian).
 *   qp [B][n] reduced coordinates; x_min, dx [n] input scaling (xs = 2 (qp - x_min)/dx - 1);
 *   XtT [n][Ns] training centres, transposed;  eps shape parameter
 *   phi [B][Ns]     <- k(|xs - Xt_i|)                                            (NULL = skip)
 *   GT  [B][n][Ns]  <- d phi_i / d qp_k  (chain rule through the input scaling included) (NULL = skip)
 * The closure value and Jacobian follow as phi . W and GT . W (one GEMM each). */
int bg_rbf_eval(int B, int n, int Ns, int kind, double eps, const double *qp, const double *x_min,
                const double *dx, const double *XtT, double *phi, double *GT, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BURGERS_HIP_H */
