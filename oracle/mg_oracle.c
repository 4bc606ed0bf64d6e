/*
 * mg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the 3-D multigrid V-cycle that
 * jcoady/MultiGrid runs inside MultiGridSolver{,V2,V3,V4,V5}::solve()
 * (cpp/PotentialSolver.cpp:2390-4248).  The five reference classes are one
 * algorithm hand-unrolled per depth; here the depth L and the sweep schedule
 * nu[0..L] are parameters.  Every function cites the reference lines it follows
 * and keeps the reference's fp64 association order, its true divisions and its
 * index conventions (coarse size n>>1, coarse smoother range 1..n-2, eps arrays
 * one node larger than R arrays, injection on the low faces only, eps fields
 * never re-zeroed), so it reproduces the reference's iterates, not merely its
 * fixed point.  Build with contraction off (no FMA) -- see oracle/Makefile.
 *
 * It also holds the red-black ordering of the same smoother ("rb"): identical
 * update formula and omega, colour (i+j+k)&1 == 0 first, then colour 1.  That is
 * the ordering the HIP library uses; the lexicographic ordering is the
 * reference's.
 *
 * PARITY STATUS: parity unpinned by reference-held fixtures.  The reference
 * ships no tests, golden files or fixtures for this path, and it cannot be
 * compiled in the build image (cpp/PotentialSolver.h:6 includes <fftw3.h>,
 * which is not installed, and stand-in headers are not allowed).  The oracle is
 * cross-checked in tests/test_oracle_known_answers.py against the residual-norm
 * histories that the structural survey recorded from the compiled reference
 * (7 significant digits, 41^3 V2 / 129^3 V3 / 257^3 V5 / 513^3 V5), and in
 * tests/test_oracle_second_transcription.py against a second transcription of
 * the reference's loops made independently of this file (plain Python, V1..V5,
 * non-cubic grids): bit for bit in phi, every R_l and eps_l and every norm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file; nothing in the product path links or calls it.
 *
 * Layout: flat row-major arrays, index ((i*nj)+j)*nk+k, k contiguous -- the
 * memory order of the reference's Field_<double> rows (cpp/Field.h:65-70).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS_0 8.85418782e-12 /* cpp/World.h:14 */
#define ORC_MAX_LEVELS 12

#define IDX(i, j, k, nj, nk) ((((int64_t)(i)) * (nj) + (j)) * (nk) + (k))

/* Coefficients of level l: 1/(4^l dx dx) written the reference's way,
 * 1.0/(f*dh*dh) with f = 4,16,64,256,1024 (cpp/PotentialSolver.cpp:3728-3750);
 * d = 2*cx+2*cy+2*cz, left-associated (e.g. :3783). c[3] = cx,cy,cz; c[3] = d. */
void orc_coeffs(const double dh[3], int level, double c[4])
{
    double f = 1.0;
    for (int l = 0; l < level; l++) f *= 4.0;
    for (int a = 0; a < 3; a++) {
        if (level == 0) c[a] = 1.0 / (dh[a] * dh[a]);
        else            c[a] = 1.0 / (f * dh[a] * dh[a]);
    }
    c[3] = 2 * c[0] + 2 * c[1] + 2 * c[2];
}

/* One smoother point.  Fine level (cpp/PotentialSolver.cpp:3780-3786):
 *   g = (rho/EPS_0 + cx*(u[i-1]+u[i+1]) + cy*(u[j-1]+u[j+1]) + cz*(u[k-1]+u[k+1]))/d
 *   u = u + 1.4*(g-u)
 * Coarse levels (:3929-3935) use R_l itself in place of rho/EPS_0. */
static inline void smooth_point(double *u, const double *rhs, int div_eps0,
                                int64_t iu, int64_t ir, int64_t si, int64_t sj,
                                const double c[4], double omega)
{
    double b = div_eps0 ? rhs[ir] / ORC_EPS_0 : rhs[ir];
    double g = (b + c[0] * (u[iu - si] + u[iu + si]) +
                c[1] * (u[iu - sj] + u[iu + sj]) +
                c[2] * (u[iu - 1] + u[iu + 1])) / c[3];
    u[iu] = u[iu] + omega * (g - u[iu]);
}

/* Lexicographic GS-SOR sweeps over 1..n-2 (i outer, k inner), in place.
 * u has dims un[3], rhs has dims rn[3]; n[3] is the level size that bounds the
 * loops (fine: n = un = rn; coarse level l: n = rn = n_l, un = n_l+1).
 * Reference: fine :2428-2443/:3773-3789, coarse :3922-3937 etc. */
void orc_smooth_lex(double *u, const int64_t un[3], const double *rhs,
                    const int64_t rn[3], const int64_t n[3], const double c[4],
                    double omega, int sweeps, int div_eps0)
{
    const int64_t si = un[1] * un[2], sj = un[2];
    for (int s = 0; s < sweeps; s++)
        for (int64_t i = 1; i < n[0] - 1; i++)
            for (int64_t j = 1; j < n[1] - 1; j++)
                for (int64_t k = 1; k < n[2] - 1; k++)
                    smooth_point(u, rhs, div_eps0, IDX(i, j, k, un[1], un[2]),
                                 IDX(i, j, k, rn[1], rn[2]), si, sj, c, omega);
}

/* Red-black ordering of the same update: all points with (i+j+k)&1 == 0, then
 * all with (i+j+k)&1 == 1, per sweep.  Within a colour the updates are
 * independent, so this loop order defines the result uniquely. */
void orc_smooth_rb(double *u, const int64_t un[3], const double *rhs,
                   const int64_t rn[3], const int64_t n[3], const double c[4],
                   double omega, int sweeps, int div_eps0)
{
    const int64_t si = un[1] * un[2], sj = un[2];
    for (int s = 0; s < sweeps; s++)
        for (int colour = 0; colour < 2; colour++)
            for (int64_t i = 1; i < n[0] - 1; i++)
                for (int64_t j = 1; j < n[1] - 1; j++)
                    for (int64_t k = 1 + ((i + j + 1 + colour) & 1); k < n[2] - 1; k += 2)
                        smooth_point(u, rhs, div_eps0, IDX(i, j, k, un[1], un[2]),
                                     IDX(i, j, k, rn[1], rn[2]), si, sj, c, omega);
}

/* The red-black sweep with the point update in its short form (the HIP library's
 * MG_FLAG_FAST_SMOOTHER; no counterpart in the reference, whose update is smooth_point above): the
 * division by d and omega are folded into the coefficients,
 *   wb = omega*(1/d), wx = (omega*cx)*(1/d), wy, wz alike, wu = 1 - omega
 *   u = fma(wz, u[k-1]+u[k+1], fma(wy, u[j-1]+u[j+1], fma(wx, u[i-1]+u[i+1], fma(wb, b, wu*u))))
 * Equal to smooth_point up to rounding; this twin exists so that the GPU's short form can be
 * checked bit for bit. */
static inline void smooth_point_fast(double *u, const double *rhs, int div_eps0,
                                     int64_t iu, int64_t ir, int64_t si, int64_t sj,
                                     const double w[5])
{
    double b = div_eps0 ? rhs[ir] / ORC_EPS_0 : rhs[ir];
    double t = w[4] * u[iu];
    t = fma(w[0], b, t);
    t = fma(w[1], u[iu - si] + u[iu + si], t);
    t = fma(w[2], u[iu - sj] + u[iu + sj], t);
    u[iu] = fma(w[3], u[iu - 1] + u[iu + 1], t);
}

void orc_smooth_rb_fast(double *u, const int64_t un[3], const double *rhs,
                        const int64_t rn[3], const int64_t n[3], const double c[4],
                        double omega, int sweeps, int div_eps0)
{
    const int64_t si = un[1] * un[2], sj = un[2];
    const double rd = 1.0 / c[3];
    const double w[5] = {omega * rd, (omega * c[0]) * rd, (omega * c[1]) * rd, (omega * c[2]) * rd, 1.0 - omega};
    for (int s = 0; s < sweeps; s++)
        for (int colour = 0; colour < 2; colour++)
            for (int64_t i = 1; i < n[0] - 1; i++)
                for (int64_t j = 1; j < n[1] - 1; j++)
                    for (int64_t k = 1 + ((i + j + 1 + colour) & 1); k < n[2] - 1; k += 2)
                        smooth_point_fast(u, rhs, div_eps0, IDX(i, j, k, un[1], un[2]),
                                          IDX(i, j, k, rn[1], rn[2]), si, sj, w);
}

/* R = A*phi - b on the fine mesh; on the six faces R = phi - 0
 * (cpp/PotentialSolver.cpp:2656-2669, :3792-3805).  Returns sum of R^2 over all
 * nodes in i,j,k order (:3810-3816). */
double orc_residual(const double *phi, const double *rho, double *R,
                    const int64_t n[3], const double c[4])
{
    const int64_t ni = n[0], nj = n[1], nk = n[2], si = nj * nk, sj = nk;
    for (int64_t i = 0; i < ni; i++)
        for (int64_t j = 0; j < nj; j++)
            for (int64_t k = 0; k < nk; k++) {
                int64_t p = IDX(i, j, k, nj, nk);
                if (i == 0 || i == ni - 1 || j == 0 || j == nj - 1 || k == 0 || k == nk - 1)
                    R[p] = phi[p] - 0;
                else
                    R[p] = phi[p] * c[3] - rho[p] / ORC_EPS_0 -
                           c[0] * (phi[p - si] + phi[p + si]) -
                           c[1] * (phi[p - sj] + phi[p + sj]) -
                           c[2] * (phi[p - 1] + phi[p + 1]);
            }
    double sum = 0;
    for (int64_t p = 0; p < ni * nj * nk; p++) sum += R[p] * R[p];
    return sum;
}

/* Sequential sum of squares (the reference's norm loop, :3810-3816). */
double orc_sumsq(const double *x, int64_t count)
{
    double sum = 0;
    for (int64_t p = 0; p < count; p++) sum += x[p] * x[p];
    return sum;
}

/* 27-point full weighting R_f (dims nf) -> R_c (dims nc = nf>>1), for
 * I,J,K = 0,2,4,... < nf-1; injection where I==0 || J==0 || K==0.
 * The 27 terms are summed left to right with the first index fastest, then the
 * second, the third slowest, then divided by 64.0
 * (cpp/PotentialSolver.cpp:2474-2491, :3822-3919). */
void orc_restrict27(const double *Rf, const int64_t nf[3], double *Rc, const int64_t nc[3])
{
    static const double w1[3] = {1, 2, 1};
    const int64_t fj = nf[1], fk = nf[2];
    for (int64_t i = 0; i < nf[0] - 1; i += 2)
        for (int64_t j = 0; j < nf[1] - 1; j += 2)
            for (int64_t k = 0; k < nf[2] - 1; k += 2) {
                double v;
                if (i == 0 || j == 0 || k == 0) {
                    v = Rf[IDX(i, j, k, fj, fk)];
                } else {
                    v = 0;
                    int first = 1;
                    for (int dk = -1; dk <= 1; dk++)
                        for (int dj = -1; dj <= 1; dj++)
                            for (int di = -1; di <= 1; di++) {
                                double w = w1[di + 1] * w1[dj + 1] * w1[dk + 1];
                                double r = Rf[IDX(i + di, j + dj, k + dk, fj, fk)];
                                /* the reference writes a bare R for weight 1 and w*R
                                 * otherwise; 1*R == R bit for bit */
                                double t = (w == 1.0) ? r : w * r;
                                if (first) { v = t; first = 0; } else v = v + t;
                            }
                    v = v / 64.0;
                }
                Rc[IDX(i / 2, j / 2, k / 2, nc[1], nc[2])] = v;
            }
}

/* The "B" solver family's restriction (MultiGridSolverB..V5B, e.g.
 * cpp/PotentialSolver.cpp:704-715, :1991-2002): same loop bounds and injection rule as above,
 * interior value = (R[i-1]+2R+R[i+1] + R[j-1]+2R+R[j+1] + R[k-1]+2R+R[k+1]) / 12.0, summed
 * left to right. */
void orc_restrict7(const double *Rf, const int64_t nf[3], double *Rc, const int64_t nc[3])
{
    const int64_t si = nf[1] * nf[2], sj = nf[2];
    for (int64_t i = 0; i < nf[0] - 1; i += 2)
        for (int64_t j = 0; j < nf[1] - 1; j += 2)
            for (int64_t k = 0; k < nf[2] - 1; k += 2) {
                const int64_t p = IDX(i, j, k, nf[1], nf[2]);
                const int64_t q = IDX(i / 2, j / 2, k / 2, nc[1], nc[2]);
                if (i == 0 || j == 0 || k == 0)
                    Rc[q] = Rf[p];
                else
                    Rc[q] = (Rf[p - si] + 2 * Rf[p] + Rf[p + si] +
                             Rf[p - sj] + 2 * Rf[p] + Rf[p + sj] +
                             Rf[p - 1] + 2 * Rf[p] + Rf[p + 1]) / 12.0;
            }
}

/* Trilinear prolongation e_c (dims cn) -> e_f (dims fn), OVERWRITING e_f on
 * 0..n-2 per axis where n[3] is the fine level's size; zero where any index is
 * 0; everything outside 0..n-2 is left untouched
 * (cpp/PotentialSolver.cpp:2513-2549, :3940-3976, ..., :4170-4206). */
void orc_prolong(const double *ec, const int64_t cn[3], double *ef,
                 const int64_t fn[3], const int64_t n[3])
{
    const int64_t cj = cn[1], ck = cn[2];
#define C(a, b, d) ec[IDX(I + (a), J + (b), K + (d), cj, ck)]
    for (int64_t i = 0; i < n[0] - 1; i++)
        for (int64_t j = 0; j < n[1] - 1; j++)
            for (int64_t k = 0; k < n[2] - 1; k++) {
                const int64_t I = i / 2, J = j / 2, K = k / 2;
                const int oi = (int)(i & 1), oj = (int)(j & 1), ok = (int)(k & 1);
                double v;
                if (i == 0 || j == 0 || k == 0) v = 0;
                else if (!oi && !oj && !ok) v = C(0, 0, 0);
                else if (oi && !oj && !ok) v = 0.5 * (C(0, 0, 0) + C(1, 0, 0));
                else if (!oi && oj && !ok) v = 0.5 * (C(0, 0, 0) + C(0, 1, 0));
                else if (!oi && !oj && ok) v = 0.5 * (C(0, 0, 0) + C(0, 0, 1));
                else if (oi && oj && !ok) v = 0.25 * (C(0, 0, 0) + C(1, 0, 0) + C(0, 1, 0) + C(1, 1, 0));
                else if (oi && !oj && ok) v = 0.25 * (C(0, 0, 0) + C(1, 0, 0) + C(0, 0, 1) + C(1, 0, 1));
                else if (!oi && oj && ok) v = 0.25 * (C(0, 0, 0) + C(0, 1, 0) + C(0, 0, 1) + C(0, 1, 1));
                else v = 0.125 * (C(0, 0, 0) + C(1, 0, 0) + C(0, 1, 0) + C(1, 1, 0) +
                                  C(0, 0, 1) + C(1, 0, 1) + C(0, 1, 1) + C(1, 1, 1));
                ef[IDX(i, j, k, fn[1], fn[2])] = v;
            }
#undef C
}

/* phi -= eps_h over the whole fine array (cpp/PotentialSolver.cpp:4209-4214). */
void orc_correct(double *phi, const double *eps, int64_t count)
{
    for (int64_t p = 0; p < count; p++) phi[p] = phi[p] - eps[p];
}

/* E = -grad(phi): central differences inside, 3-point one-sided on the faces
 * (cpp/PotentialSolver.cpp:418-458).  ef is [ni][nj][nk][3]. */
void orc_compute_ef(const double *phi, double *ef, const int64_t n[3], const double dh[3])
{
    const int64_t ni = n[0], nj = n[1], nk = n[2], si = nj * nk, sj = nk;
    const double dx = dh[0], dy = dh[1], dz = dh[2];
    for (int64_t i = 0; i < ni; i++)
        for (int64_t j = 0; j < nj; j++)
            for (int64_t k = 0; k < nk; k++) {
                int64_t p = IDX(i, j, k, nj, nk);
                double *e = ef + 3 * p;
                if (i == 0) e[0] = -(-3 * phi[p] + 4 * phi[p + si] - phi[p + 2 * si]) / (2 * dx);
                else if (i == ni - 1) e[0] = -(phi[p - 2 * si] - 4 * phi[p - si] + 3 * phi[p]) / (2 * dx);
                else e[0] = -(phi[p + si] - phi[p - si]) / (2 * dx);
                if (j == 0) e[1] = -(-3 * phi[p] + 4 * phi[p + sj] - phi[p + 2 * sj]) / (2 * dy);
                else if (j == nj - 1) e[1] = -(phi[p - 2 * sj] - 4 * phi[p - sj] + 3 * phi[p]) / (2 * dy);
                else e[1] = -(phi[p + sj] - phi[p - sj]) / (2 * dy);
                if (k == 0) e[2] = -(-3 * phi[p] + 4 * phi[p + 1] - phi[p + 2]) / (2 * dz);
                else if (k == nk - 1) e[2] = -(phi[p - 2] - 4 * phi[p - 1] + 3 * phi[p]) / (2 * dz);
                else e[2] = -(phi[p + 1] - phi[p - 1]) / (2 * dz);
            }
}

/* GaussSeidelSolver::solve (cpp/PotentialSolver.cpp:235-291): one SOR sweep per iteration
 * (lexicographic as the reference, or red-black when rb); in iterations with it%25 == 0 the
 * norm sqrt(sum/(ni*nj*nk)) of R = -phi*d + rho/EPS_0 + cx*(..) + cy*(..) + cz*(..) summed over
 * the INTERIOR nodes in i,j,k order is tested (:268-286).  Returns converged; *iterations =
 * sweeps performed, *l2_out = last norm evaluated. */
int orc_gs_solve(double *phi, const double *rho, const int64_t n[3], const double dh[3],
                 unsigned max_it, double tol, double omega, int rb, unsigned *iterations, double *l2_out)
{
    const int64_t ni = n[0], nj = n[1], nk = n[2], si = nj * nk, sj = nk;
    double c[4];
    orc_coeffs(dh, 0, c);
    double L2 = 0;
    int converged = 0;
    unsigned it;
    for (it = 0; it < max_it; it++) {
        if (rb) orc_smooth_rb(phi, n, rho, n, n, c, omega, 1, 1);
        else orc_smooth_lex(phi, n, rho, n, n, c, omega, 1, 1);
        if (it % 25 == 0) {
            double sum = 0;
            for (int64_t i = 1; i < ni - 1; i++)
                for (int64_t j = 1; j < nj - 1; j++)
                    for (int64_t k = 1; k < nk - 1; k++) {
                        int64_t p = IDX(i, j, k, nj, nk);
                        double R = -phi[p] * c[3] + rho[p] / ORC_EPS_0 +
                                   c[0] * (phi[p - si] + phi[p + si]) +
                                   c[1] * (phi[p - sj] + phi[p + sj]) +
                                   c[2] * (phi[p - 1] + phi[p + 1]);
                        sum += R * R;
                    }
            L2 = sqrt(sum / (double)(ni * nj * nk));
            if (L2 < tol) { converged = 1; it++; break; }
        }
    }
    if (iterations) *iterations = it;
    if (l2_out) *l2_out = L2;
    return converged;
}

/* ------------------------------------------------------------------------- */
/* Whole solver: the state the reference keeps in class members
 * (cpp/PotentialSolver.h:171-272): R_l (n_l^3), eps_0 (n_0^3), eps_l ((n_l+1)^3),
 * all zero at construction and never cleared afterwards. */
typedef struct {
    int depth;                       /* L = number of coarse grids */
    int restrict7;                   /* 1: the "B" classes' 7-point restriction */
    int nu[ORC_MAX_LEVELS];          /* sweeps per level, nu[0] on the fine grid */
    double omega;
    double dh[3];
    int64_t n[ORC_MAX_LEVELS][3];    /* n_l = n_0 >> l */
    int64_t en[ORC_MAX_LEVELS][3];   /* dims of eps_l */
    double *R[ORC_MAX_LEVELS];
    double *eps[ORC_MAX_LEVELS];
} orc_solver;

void orc_solver_free(orc_solver *s)
{
    if (!s) return;
    for (int l = 0; l < ORC_MAX_LEVELS; l++) { free(s->R[l]); free(s->eps[l]); }
    free(s);
}

orc_solver *orc_solver_new(const int64_t n0[3], const double dh[3], int depth,
                           const int *nu, double omega)
{
    if (depth < 1 || depth >= ORC_MAX_LEVELS) return NULL;
    orc_solver *s = (orc_solver *)calloc(1, sizeof(orc_solver));
    if (!s) return NULL;
    s->depth = depth;
    s->omega = omega;
    for (int a = 0; a < 3; a++) s->dh[a] = dh[a];
    for (int l = 0; l <= depth; l++) {
        s->nu[l] = nu[l];
        int64_t cnt = 1, ecnt = 1;
        for (int a = 0; a < 3; a++) {
            s->n[l][a] = n0[a] >> l;
            s->en[l][a] = (l == 0) ? s->n[l][a] : s->n[l][a] + 1;
            cnt *= s->n[l][a];
            ecnt *= s->en[l][a];
        }
        if (s->n[l][0] < 3 || s->n[l][1] < 3 || s->n[l][2] < 3) { orc_solver_free(s); return NULL; }
        s->R[l] = (double *)calloc((size_t)cnt, sizeof(double));
        s->eps[l] = (double *)calloc((size_t)ecnt, sizeof(double));
        if (!s->R[l] || !s->eps[l]) { orc_solver_free(s); return NULL; }
    }
    return s;
}

void orc_solver_reset(orc_solver *s)
{
    for (int l = 0; l <= s->depth; l++) {
        memset(s->R[l], 0, sizeof(double) * (size_t)(s->n[l][0] * s->n[l][1] * s->n[l][2]));
        memset(s->eps[l], 0, sizeof(double) * (size_t)(s->en[l][0] * s->en[l][1] * s->en[l][2]));
    }
}

void orc_solver_set_restrict7(orc_solver *s, int on) { s->restrict7 = on != 0; }
double *orc_solver_R(orc_solver *s, int l) { return s->R[l]; }
double *orc_solver_eps(orc_solver *s, int l) { return s->eps[l]; }

typedef void (*smooth_fn)(double *, const int64_t *, const double *, const int64_t *,
                          const int64_t *, const double *, double, int, int);

/* The iteration loop of solve() (cpp/PotentialSolver.cpp:3756-4216 for V5; the
 * other depths are the same text with fewer levels): smooth, residual, norm and
 * termination test, restrict down, then (smooth, prolong) up, correct.
 * rb != 0 selects the red-black ordering for every smoother call (rb == 2: on the levels >= 1
 * with the point update in its short form, orc_smooth_rb_fast, as the HIP library's
 * MG_FLAG_FAST_SMOOTHER does).
 * l2_hist (may be NULL) receives the norm observed in each iteration.
 * Returns 1 when L2 < tol was observed, else 0; *cycles = iterations whose
 * residual was evaluated (the converged one included). */
int orc_solve(orc_solver *s, double *phi, const double *rho, unsigned max_it,
              double tol, int rb, unsigned *cycles, double *L2_out, double *l2_hist)
{
    smooth_fn smooth = rb ? orc_smooth_rb : orc_smooth_lex;              /* fine grid */
    smooth_fn smooth_c = rb == 2 ? orc_smooth_rb_fast : smooth;          /* levels >= 1 */
    const int L = s->depth;
    double c[ORC_MAX_LEVELS][4];
    for (int l = 0; l <= L; l++) orc_coeffs(s->dh, l, c[l]);
    const int64_t *n0 = s->n[0];
    const int64_t N0 = n0[0] * n0[1] * n0[2];
    double L2 = 0;
    int converged = 0;
    unsigned it;
    for (it = 0; it < max_it; it++) {
        smooth(phi, n0, rho, n0, n0, c[0], s->omega, s->nu[0], 1);
        double sum = orc_residual(phi, rho, s->R[0], n0, c[0]);
        L2 = sqrt(sum / (double)N0);
        if (l2_hist) l2_hist[it] = L2;
        if (L2 < tol) { converged = 1; it++; break; }
        for (int l = 0; l < L; l++) {
            if (s->restrict7) orc_restrict7(s->R[l], s->n[l], s->R[l + 1], s->n[l + 1]);
            else orc_restrict27(s->R[l], s->n[l], s->R[l + 1], s->n[l + 1]);
        }
        smooth_c(s->eps[L], s->en[L], s->R[L], s->n[L], s->n[L], c[L], s->omega, s->nu[L], 0);
        for (int l = L - 1; l >= 1; l--) {
            orc_prolong(s->eps[l + 1], s->en[l + 1], s->eps[l], s->en[l], s->n[l]);
            smooth_c(s->eps[l], s->en[l], s->R[l], s->n[l], s->n[l], c[l], s->omega, s->nu[l], 0);
        }
        orc_prolong(s->eps[1], s->en[1], s->eps[0], s->en[0], s->n[0]);
        orc_correct(phi, s->eps[0], N0);
    }
    if (cycles) *cycles = it;
    if (L2_out) *L2_out = L2;
    return converged;
}

/* ------------------------------------------------------------------------- */
/* Slab forms: the same loop nests on a SLAB of a level's global array -- planes
 * [i0, i0+d[0]) stored with row pitch sj -- producing a range [lo, hi) of global
 * planes.  They are the CPU twin of the device-level operators the slab-partitioned
 * (multi-GPU) solve is built from, and reduce to the functions above for a slab
 * that is the whole array.  The reference has no counterpart (it is a single
 * process); what is restated is still its arithmetic, per point. */
typedef struct {
    double *p;
    int64_t d[3];   /* stored planes, rows, columns */
    int64_t si, sj; /* strides in doubles */
    int64_t i0;     /* global index of stored plane 0 */
} orc_view;

#define VAT(v, i, j, k) ((v)->p[((i) - (v)->i0) * (v)->si + (j) * (v)->sj + (k)])

/* One red-black sweep of planes [lo,hi): red (colour 0) also on planes lo-1 and hi where
 * their neighbours are stored, then black on [lo,hi); updates stay within 1..n-2. */
void orc_slab_smooth_rb(orc_view *u, const orc_view *rhs, const int64_t n[3], const double c[4],
                        double omega, int64_t lo, int64_t hi)
{
    for (int colour = 0; colour < 2; colour++) {
        int64_t a = colour == 0 ? lo - 1 : lo, b = colour == 0 ? hi + 1 : hi;
        if (colour == 0) {
            if (a < u->i0 + 1) a = u->i0 + 1;
            if (b > u->i0 + u->d[0] - 1) b = u->i0 + u->d[0] - 1;
        }
        if (a < 1) a = 1;
        if (b > n[0] - 1) b = n[0] - 1;
        for (int64_t i = a; i < b; i++)
            for (int64_t j = 1; j < n[1] - 1; j++)
                for (int64_t k = 1 + ((i + j + 1 + colour) & 1); k < n[2] - 1; k += 2) {
                    double g = (VAT(rhs, i, j, k) + c[0] * (VAT(u, i - 1, j, k) + VAT(u, i + 1, j, k)) +
                                c[1] * (VAT(u, i, j - 1, k) + VAT(u, i, j + 1, k)) +
                                c[2] * (VAT(u, i, j, k - 1) + VAT(u, i, j, k + 1))) / c[3];
                    VAT(u, i, j, k) = VAT(u, i, j, k) + omega * (g - VAT(u, i, j, k));
                }
    }
}

static double slab_residual_at(const orc_view *phi, const orc_view *b, const int64_t n[3],
                               const double c[4], int64_t i, int64_t j, int64_t k)
{
    if (i == 0 || i == n[0] - 1 || j == 0 || j == n[1] - 1 || k == 0 || k == n[2] - 1)
        return VAT(phi, i, j, k) - 0;
    return VAT(phi, i, j, k) * c[3] - VAT(b, i, j, k) -
           c[0] * (VAT(phi, i - 1, j, k) + VAT(phi, i + 1, j, k)) -
           c[1] * (VAT(phi, i, j - 1, k) + VAT(phi, i, j + 1, k)) -
           c[2] * (VAT(phi, i, j, k - 1) + VAT(phi, i, j, k + 1));
}

/* Coarse planes I with lo <= 2I < hi of Rc = restrict27(residual(phi, b)); b = rho/EPS_0.
 * Returns the sum of squares of the fine residual over planes [lo,hi), i,j,k order. */
double orc_slab_residual_restrict(const orc_view *phi, const orc_view *b, orc_view *Rc,
                                  const int64_t n[3], const double c[4], int64_t lo, int64_t hi)
{
    static const double w1[3] = {1, 2, 1};
    double sum = 0;
    for (int64_t i = lo; i < hi; i++)
        for (int64_t j = 0; j < n[1]; j++)
            for (int64_t k = 0; k < n[2]; k++) {
                double r = slab_residual_at(phi, b, n, c, i, j, k);
                sum += r * r;
            }
    for (int64_t i = lo; i < hi && i < n[0] - 1; i += 2)
        for (int64_t j = 0; j < n[1] - 1; j += 2)
            for (int64_t k = 0; k < n[2] - 1; k += 2) {
                double v;
                if (i == 0 || j == 0 || k == 0) {
                    v = slab_residual_at(phi, b, n, c, i, j, k);
                } else {
                    int first = 1;
                    v = 0;
                    for (int dk = -1; dk <= 1; dk++)
                        for (int dj = -1; dj <= 1; dj++)
                            for (int di = -1; di <= 1; di++) {
                                double w = w1[di + 1] * w1[dj + 1] * w1[dk + 1];
                                double r = slab_residual_at(phi, b, n, c, i + di, j + dj, k + dk);
                                double t = (w == 1.0) ? r : w * r;
                                if (first) { v = t; first = 0; } else v = v + t;
                            }
                    v = v / 64.0;
                }
                VAT(Rc, i / 2, j / 2, k / 2) = v;
            }
    return sum;
}

/* Coarse planes [lo,hi) of Rc = restrict27(Rf); nf = fine level size. */
void orc_slab_restrict27(const orc_view *Rf, orc_view *Rc, const int64_t nf[3], int64_t lo, int64_t hi)
{
    static const double w1[3] = {1, 2, 1};
    for (int64_t I = lo; I < hi; I++) {
        const int64_t i = 2 * I;
        if (i >= nf[0] - 1) break;
        for (int64_t j = 0; j < nf[1] - 1; j += 2)
            for (int64_t k = 0; k < nf[2] - 1; k += 2) {
                double v;
                if (i == 0 || j == 0 || k == 0) {
                    v = VAT(Rf, i, j, k);
                } else {
                    int first = 1;
                    v = 0;
                    for (int dk = -1; dk <= 1; dk++)
                        for (int dj = -1; dj <= 1; dj++)
                            for (int di = -1; di <= 1; di++) {
                                double w = w1[di + 1] * w1[dj + 1] * w1[dk + 1];
                                double r = VAT(Rf, i + di, j + dj, k + dk);
                                double t = (w == 1.0) ? r : w * r;
                                if (first) { v = t; first = 0; } else v = v + t;
                            }
                    v = v / 64.0;
                }
                VAT(Rc, I, j / 2, k / 2) = v;
            }
    }
}

/* Fine planes [lo,hi) within 0..n-2: out = P(ec), or out -= P(ec) when correct. */
void orc_slab_prolong(const orc_view *ec, orc_view *out, const int64_t n[3], int64_t lo, int64_t hi,
                      int correct)
{
#define C(a, b, d) VAT(ec, I + (a), J + (b), K + (d))
    for (int64_t i = lo; i < hi && i < n[0] - 1; i++)
        for (int64_t j = 0; j < n[1] - 1; j++)
            for (int64_t k = 0; k < n[2] - 1; k++) {
                const int64_t I = i / 2, J = j / 2, K = k / 2;
                const int oi = (int)(i & 1), oj = (int)(j & 1), ok = (int)(k & 1);
                double v;
                if (i == 0 || j == 0 || k == 0) v = 0;
                else if (!oi && !oj && !ok) v = C(0, 0, 0);
                else if (oi && !oj && !ok) v = 0.5 * (C(0, 0, 0) + C(1, 0, 0));
                else if (!oi && oj && !ok) v = 0.5 * (C(0, 0, 0) + C(0, 1, 0));
                else if (!oi && !oj && ok) v = 0.5 * (C(0, 0, 0) + C(0, 0, 1));
                else if (oi && oj && !ok) v = 0.25 * (C(0, 0, 0) + C(1, 0, 0) + C(0, 1, 0) + C(1, 1, 0));
                else if (oi && !oj && ok) v = 0.25 * (C(0, 0, 0) + C(1, 0, 0) + C(0, 0, 1) + C(1, 0, 1));
                else if (!oi && oj && ok) v = 0.25 * (C(0, 0, 0) + C(0, 1, 0) + C(0, 0, 1) + C(0, 1, 1));
                else v = 0.125 * (C(0, 0, 0) + C(1, 0, 0) + C(0, 1, 0) + C(1, 1, 0) +
                                  C(0, 0, 1) + C(1, 0, 1) + C(0, 1, 1) + C(1, 1, 1));
                if (correct) VAT(out, i, j, k) = VAT(out, i, j, k) - v;
                else VAT(out, i, j, k) = v;
            }
#undef C
}

/* v /= EPS_0 over the stored slab (rows of d[2] nodes). */
void orc_slab_div_eps0(orc_view *v)
{
    for (int64_t i = 0; i < v->d[0]; i++)
        for (int64_t j = 0; j < v->d[1]; j++)
            for (int64_t k = 0; k < v->d[2]; k++) v->p[i * v->si + j * v->sj + k] /= ORC_EPS_0;
}
