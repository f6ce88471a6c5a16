/* CPU oracle, C restatement of the FOM hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product (1d-burgers-equation-roms_amd/) never links or loads it.
 *
 * It restates, in closed form for 2-node P1 elements, one Picard iteration of
 * the reference's FEMBurgers.fom_burgers (FEM/fem_burgers.py:646-707):
 *   convection matrix   FEM/fem_burgers.py:389-425
 *   forcing vector      FEM/fem_burgers.py:427-461
 *   SUPG vector         FEM/fem_burgers.py:500-581
 *   A = M + dt*C + dt*E*K, Dirichlet row 0, b, R, solve, update  :676-698
 * The closed forms are those of SURVEY.md Appendix A.  The linear solve is a
 * partially pivoted tridiagonal elimination (LAPACK dgtsv's algorithm), which
 * is what the reference's SuperLU call amounts to on a tridiagonal matrix.
 *
 * Parity: PINNED through tests/test_oracle_golden.py, which checks this
 * library against oracle/burgers_ref.py and against the golden fixtures
 * (reference-run vectors and slices of the reference's committed .npy files).
 *
 * Build: see oracle/Makefile  (gcc -O3 -fopenmp -shared -fPIC; no -march: the .so travels to the GPU box).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GP1 0.78867513459481287  /* (1 + 1/sqrt 3)/2 */
#define GP2 0.21132486540518713  /* (1 - 1/sqrt 3)/2 */

/* Pivoted tridiagonal solve, dgtsv-style.  dl[0..n-2], d[0..n-1], du[0..n-2]
 * are destroyed; b is overwritten with the solution.  Returns 0 or the
 * (1-based) index of a zero pivot. */
static int gtsv(int n, double *dl, double *d, double *du, double *du2, double *b)
{
    for (int i = 0; i < n - 1; ++i) {
        if (fabs(d[i]) >= fabs(dl[i])) {
            if (d[i] == 0.0) return i + 1;
            double f = dl[i] / d[i];
            d[i + 1] -= f * du[i];
            b[i + 1] -= f * b[i];
            du2[i] = 0.0;
        } else {
            double f = d[i] / dl[i];
            d[i] = dl[i];
            double t = d[i + 1];
            d[i + 1] = du[i] - f * t;
            if (i < n - 2) { du2[i] = du[i + 1]; du[i + 1] = -f * du[i + 1]; } else du2[i] = 0.0;
            du[i] = t;
            t = b[i]; b[i] = b[i + 1]; b[i + 1] = t - f * b[i + 1];
        }
    }
    if (d[n - 1] == 0.0) return n;
    b[n - 1] /= d[n - 1];
    if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / d[n - 2];
    for (int i = n - 3; i >= 0; --i)
        b[i] = (b[i] - du[i] * b[i + 1] - du2[i] * b[i + 2]) / d[i];
    return 0;
}

/* Per-sample constants: F (forcing load vector) and fs_e = f(gp1)+f(gp2). */
static void forcing(int n, const double *X, double mu2, double *F, double *fs)
{
    memset(F, 0, sizeof(double) * n);
    for (int e = 0; e < n - 1; ++e) {
        double xl = X[e], xr = X[e + 1], h = xr - xl;
        double f1 = 0.02 * exp(mu2 * (GP1 * xl + GP2 * xr));
        double f2 = 0.02 * exp(mu2 * (GP2 * xl + GP1 * xr));
        F[e]     += (f1 * GP1 + f2 * GP2) * (0.5 * h);
        F[e + 1] += (f1 * GP2 + f2 * GP1) * (0.5 * h);
        fs[e] = f1 + f2;
    }
}

/* One sample, whole time loop.  hist is [nsteps+1][n] (time-major).  Loop-invariant element
 * constants (h/3, h/6, dt*E/h) are hoisted out of the Picard loop; their values are unchanged. */
static void fom_one(int n, int nsteps, const double *X, const double *u0, double mu1, double mu2,
                    double dt, double E, double tol, int max_it, int supg,
                    double *hist, int *iters, double *w)
{
    double *F = w, *fs = w + n, *g = w + 2 * n, *lo = w + 3 * n, *di = w + 4 * n, *up = w + 5 * n,
           *du2 = w + 6 * n, *r = w + 7 * n, *u = w + 8 * n, *S = w + 9 * n, *h3 = w + 10 * n,
           *h6 = w + 11 * n, *eh = w + 12 * n, *hh = w + 13 * n;
    forcing(n, X, mu2, F, fs);
    for (int e = 0; e < n - 1; ++e) {
        double h = X[e + 1] - X[e];
        hh[e] = h; h3[e] = h / 3.0; h6[e] = h / 6.0; eh[e] = dt * E / h;
    }
    memcpy(hist, u0, sizeof(double) * n);
    for (int s = 0; s < nsteps; ++s) {
        const double *un = hist + (size_t)s * n;
        double *unew = hist + (size_t)(s + 1) * n;
        /* g = M u^n + dt F   (constant over the Picard iterations) */
        for (int i = 0; i < n; ++i) g[i] = 0.0;
        for (int e = 0; e < n - 1; ++e) {
            g[e]     += h6[e] * (2.0 * un[e] + un[e + 1]);
            g[e + 1] += h6[e] * (un[e] + 2.0 * un[e + 1]);
        }
        for (int i = 0; i < n; ++i) g[i] += dt * F[i];
        memcpy(u, un, sizeof(double) * n);
        double err = 1.0;
        int k = 0;
        while (err > tol && k < max_it) {
            for (int i = 0; i < n; ++i) { di[i] = 0.0; S[i] = 0.0; }
            lo[0] = 0.0; up[n - 1] = 0.0;
            for (int e = 0; e < n - 1; ++e) {
                double ul = u[e], ur = u[e + 1];
                double c1 = (2.0 * ul + ur) / 6.0, c2 = (ul + 2.0 * ur) / 6.0;
                di[e]     += h3[e] - dt * c1 + eh[e];
                up[e]      = h6[e] + dt * c1 - eh[e];     /* A[e, e+1] */
                lo[e + 1]  = h6[e] - dt * c2 - eh[e];     /* A[e+1, e] */
                di[e + 1] += h3[e] + dt * c2 + eh[e];
                if (supg) {
                    double h = hh[e];
                    double ub = 0.5 * (ul + ur);
                    double vel = fabs(ub) > 1e-10 ? fabs(ub) : 1e-10;
                    double tau = 0.5 * h / (2.0 * vel);
                    double se = tau * (ub * ((ur - ul) / h) - 0.5 * fs[e]);
                    S[e] -= se;
                    S[e + 1] += se;
                }
            }
            lo[0] = 0.0; di[0] = 1.0; up[0] = 0.0;      /* Dirichlet row */
            for (int i = 0; i < n; ++i) {
                double b = (i == 0) ? mu1 : g[i] - dt * S[i];
                double Au = di[i] * u[i];
                if (i > 0) Au += lo[i] * u[i - 1];
                if (i < n - 1) Au += up[i] * u[i + 1];
                r[i] = -(Au - b);
            }
            gtsv(n, lo + 1, di, up, du2, r);            /* r <- delta */
            double nd = 0.0, nu = 0.0;
            for (int i = 0; i < n; ++i) { u[i] += r[i]; nd += r[i] * r[i]; nu += u[i] * u[i]; }
            err = sqrt(nd) / sqrt(nu);
            ++k;
        }
        iters[s] = k;
        memcpy(unew, u, sizeof(double) * n);
    }
}

/* Batched driver.  u0: [B][n]; hist: [B][nsteps+1][n]; iters: [B][nsteps].
 * supg=1 for fom_burgers; nthreads<=0 -> OpenMP default. */
int bo_fom_run(int n, int B, int nsteps, const double *X, const double *u0, const double *mu1,
               const double *mu2, double dt, double E, double tol, int max_it, int supg,
               double *hist, int *iters, int nthreads)
{
    if (n < 2 || B < 0 || nsteps < 0) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int fail = 0;
#pragma omp parallel
    {
        double *w = (double *)malloc(sizeof(double) * 14 * (size_t)n);
        if (!w) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int b = 0; b < B; ++b)
                fom_one(n, nsteps, X, u0 + (size_t)b * n, mu1[b], mu2[b], dt, E, tol, max_it, supg,
                        hist + (size_t)b * (nsteps + 1) * n, iters + (size_t)b * nsteps, w);
            free(w);
        }
    }
    return fail ? -2 : 0;
}

int bo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
