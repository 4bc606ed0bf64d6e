/*
 * mg_hip.h -- C ABI of libmg_hip: the 3-D multigrid Poisson V-cycle of
 * jcoady/MultiGrid's PotentialSolver family as hand-written HIP kernels for
 * AMD MI355X (gfx950).
 *
 * The library replaces the body of MultiGridSolver{,V2,V3,V4,V5}::solve()
 * (reference cpp/PotentialSolver.cpp:2390-4248).  A host-side PotentialSolver
 * subclass packs World::phi / World::rho (Field_<double>, rows of nk doubles,
 * reference cpp/Field.h:65-70) into flat arrays with index ((i*nj)+j)*nk+k --
 * the reference's own memory order -- and calls the functions below.
 *
 * Conventions
 *   - plain C, no C++ types, nothing throws across the boundary;
 *   - every function returns MG_OK (0) or a negative mg_status; the text of the
 *     last failure on the calling thread is mg_last_error();
 *   - "flat" host arrays are caller owned, fp64, ((i*nj)+j)*nk+k, k contiguous;
 *   - a handle is bound to one HIP device and is not thread safe (the reference
 *     solver is not re-entrant either: it mutates its members); different handles
 *     may be used from different host threads at the same time;
 *   - the smoother is red-black Gauss-Seidel/SOR, colour (i+j+k)&1 == 0 first,
 *     with the reference's update formula and omega; every other operator
 *     reproduces the reference's arithmetic bit for bit (same association
 *     order, correctly rounded divisions, no FMA contraction that would drop a
 *     rounding: only products that are exact, by 2, 4 or 8, are fused).
 */
#ifndef MG_HIP_H
#define MG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 2
#define MG_MAX_LEVELS 12 /* fine grid + up to 11 coarse grids */
#define MG_MAX_GPUS 16   /* ranks of one slab-partitioned solve */

typedef enum mg_status {
    MG_OK = 0,
    MG_ERR_INVALID = -1,   /* bad argument (sizes, depth, null pointer) */
    MG_ERR_HIP = -2,       /* a HIP runtime call failed */
    MG_ERR_NOMEM = -3,     /* device or host allocation failed */
    MG_ERR_NO_DEVICE = -4, /* no usable gfx950 device */
    MG_ERR_STATE = -5,     /* call out of order (e.g. solve before set_rho) */
    MG_ERR_COMM = -6       /* RCCL failed or is unavailable, or another rank of the group failed */
} mg_status;

typedef struct mg_handle mg_handle;

/* Solver description.  Replaces the constructor arguments and hard-coded
 * constants of the reference classes:
 *   n, dh   World::ni,nj,nk and World::getDh()          (cpp/World.h:50,92)
 *   depth   number of coarse grids: 1 = MultiGridSolver, 2 = V2, ... 5 = V5
 *           (cpp/PotentialSolver.h:171-272); level l has n>>l nodes per axis.
 *           0 = no coarse grids: the handle serves mg_gs_solve / mg_compute_ef only
 *   nu      sweeps per level, nu[0] on the fine grid, nu[depth] on the coarsest
 *           (cpp/PotentialSolver.cpp:2422-2425, :2629-2634, :2922-2929,
 *            :3298-3307, :3759-3770)
 *   omega   SOR factor, 1.4 in the reference (cpp/PotentialSolver.cpp:3786)
 *   device  HIP device ordinal
 *   flags   0, or MG_FLAG_* bits
 *   n_gpus  0 or 1: the whole grid on `device`.  2..MG_MAX_GPUS: the grid is split into slabs of
 *           i-planes over devices[0..n_gpus) and solved by one host thread per GPU inside the
 *           library (halo planes over RCCL send/recv; when an ordinal repeats in devices[], over
 *           device-to-device copies, which lets the partitioned path run on one GPU).  Such a
 *           handle serves mg_set_rho / mg_set_phi / mg_solve / mg_vcycle(s) / mg_get_phi /
 *           mg_last_call_ms / mg_destroy; depth >= 1, flags 0 or MG_FLAG_FAST_SMOOTHER.
 *   coarse_cut  first level that is not partitioned: every GPU holds it and the coarser ones whole
 *           and solves them itself (0 = the first one with <= 128 planes) */
typedef struct mg_config {
    int64_t n[3];
    double dh[3];
    int32_t depth;
    int32_t nu[MG_MAX_LEVELS];
    double omega;
    int32_t device;
    int32_t flags;  /* MG_FLAG_* */
    int32_t n_gpus;
    int32_t devices[MG_MAX_GPUS];
    int32_t coarse_cut;
} mg_config;

/* Keep the fine-grid residual R_h in device memory, as the reference does
 * (cpp/PotentialSolver.h:188).  Nothing but the norm and the restriction reads
 * R_h, so by default it is produced and consumed on chip and never stored;
 * with this flag the cycle uses separate residual and restriction kernels and
 * mg_get_level_array(h, 0, 0, ...) works.  Results are bit-identical either way. */
#define MG_FLAG_KEEP_FINE_RESIDUAL 1
/* Restrict with the 7-point weights of the reference's "B" classes
 * (MultiGridSolverB..V5B, cpp/PotentialSolver.cpp:704-715, :1991-2002) instead
 * of the 27-point full weighting.  The cycle then runs the separate residual and
 * restriction kernels (R_h is stored). */
#define MG_FLAG_RESTRICT_7PT 2
/* Allocate the electric field's device array (3 doubles per node) in mg_create instead of in the
 * first mg_compute_ef, for callers that follow every solve with computeEF() (cpp/Main.cpp:124-126,
 * :157-166) and time or capture that call.  Single-GPU handles. */
#define MG_FLAG_RESERVE_EF 4
/* Coarse-level smoothers in their short form.  By default a point update evaluates the reference's
 * expression in the reference's order (cpp/PotentialSolver.cpp:3929-3935: three products, a sum, a
 * correctly rounded division by the stencil diagonal, then e + omega*(g - e)) so that the
 * red-black iterates can be compared bit for bit with a scalar CPU evaluation.  With this flag the
 * sweeps on every level l >= 1 evaluate the same update as one chain of fused multiply-adds on
 * coefficients that already carry the division and omega,
 *   e = fma(wz, e[k-1]+e[k+1], fma(wy, e[j-1]+e[j+1], fma(wx, e[i-1]+e[i+1], fma(wb, R, (1-omega)*e))))
 * with wb = omega*(1/d), wx = (omega*cx)*(1/d), ...: 8 floating-point operations instead of 22, on
 * the levels whose sweeps are bound by instruction issue.  The fine-grid sweep (bound by memory,
 * where the short form does not pay), residual, restriction, prolongation and the norm are
 * unaffected and stay exact, so a solve still stops on an exactly evaluated residual; the two
 * forms differ by rounding (a few ulp per update).  Handles of any kind (also n_gpus > 1); no
 * effect on mg_gs_solve, which sweeps the fine grid. */
#define MG_FLAG_FAST_SMOOTHER 8

/* Accumulated device time per operator, measured with HIP events on the
 * handle's stream while profiling is on (mg_set_profiling).  Index with
 * mg_timer_kind. */
typedef enum mg_timer_kind {
    MG_T_SMOOTH_FINE = 0, /* fine-level smoother launches */
    MG_T_RESIDUAL,        /* fine residual (+ sum of squares) */
    MG_T_RESTRICT,        /* all restriction launches */
    MG_T_SMOOTH_COARSE,   /* all coarse-level smoother launches */
    MG_T_PROLONG,         /* coarse-to-coarse prolongations */
    MG_T_PROLONG_CORRECT, /* fine prolongation fused with phi -= eps */
    MG_T_NORM,            /* final reduction of the partial sums */
    MG_T_EF,              /* electric field E = -grad(phi) */
    MG_T_COUNT
} mg_timer_kind;

typedef struct mg_timers {
    double ms[MG_T_COUNT];       /* summed kernel time, milliseconds */
    uint64_t launches[MG_T_COUNT];
    double bytes[MG_T_COUNT];    /* summed algorithmic bytes of those launches */
} mg_timers;

/* ---- library ---------------------------------------------------------- */
int mg_abi_version(void);
const char *mg_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int mg_device_count(void);
/* Division by the stencil diagonal (cpp/PotentialSolver.cpp:3783): by default the kernels form the
 * correctly rounded quotient from the reciprocal with two FMA corrections; on != 0 makes every
 * solver object and operator call created from now on use the division instruction instead.
 * Both give the reference's bits; the switch exists so that tests can hold one against the other. */
int mg_set_exact_division(int on);

/* ---- solver object: replaces `new MultiGridSolverVn(world, max_it, tol)` and
 * its members R_h, R_2h.., eps_h, eps_2h.. (cpp/PotentialSolver.h:188-270,
 * cpp/Main.cpp:88-100) ---------------------------------------------------- */
/* Fills cfg with the reference schedule for `depth` (1..5 as the classes;
 * deeper: 1 fine sweep, 5 per intermediate level, 50 on the coarsest). */
int mg_default_config(mg_config *cfg, int64_t ni, int64_t nj, int64_t nk,
                      const double dh[3], int depth);
/* The same for the "B" classes (cpp/PotentialSolver.h:274-350): MG_FLAG_RESTRICT_7PT,
 * and for depth 5 the schedule 1,2,3,4,5,50 (cpp/PotentialSolver.cpp:1928-1939). */
int mg_default_config_b(mg_config *cfg, int64_t ni, int64_t nj, int64_t nk,
                        const double dh[3], int depth);
int mg_create(const mg_config *cfg, mg_handle **out);
/* Device memory mg_create(cfg) will allocate, in bytes: the whole set on one GPU, or with
 * n_gpus > 1 the largest share of any GPU (its slabs plus the whole small levels).
 * Pure arithmetic, no device needed: for choosing n_gpus before anything is allocated (2049^3
 * with seven coarse grids: 236 GB on one GPU, 30 GB per GPU on eight).  The field's array of
 * mg_compute_ef (24 bytes per fine node) is counted only with MG_FLAG_RESERVE_EF.  No counterpart
 * in the reference, whose Fields live in host memory (cpp/PotentialSolver.h:171-272). */
int mg_device_bytes(const mg_config *cfg, uint64_t *bytes);
int mg_destroy(mg_handle *h);

/* world.rho -> device (stored as rho/EPS_0, the only form the path uses;
 * cpp/PotentialSolver.cpp:3780,:3800). */
int mg_set_rho(mg_handle *h, const double *rho_flat);
/* world.phi -> device: initial guess and Dirichlet face values
 * (cpp/World.cpp:40-52). */
int mg_set_phi(mg_handle *h, const double *phi_flat);
/* device -> world.phi */
int mg_get_phi(mg_handle *h, double *phi_flat);

/* The iteration loop of solve() (cpp/PotentialSolver.cpp:3756-4216): at most
 * max_it cycles; stops in the first cycle whose residual norm
 * sqrt(sum(R_h^2)/(ni*nj*nk)) is < tol, after smoothing and before that cycle's
 * coarse-grid correction, exactly like the reference.  *converged = 1/0,
 * *cycles = residual evaluations performed, *L2 = last norm.  A solve that
 * does not converge is not an error (the reference prints to cerr and returns
 * false, cpp/PotentialSolver.cpp:4246-4247); the host adapter prints. */
int mg_solve(mg_handle *h, unsigned max_it, double tol, int *converged,
             unsigned *cycles, double *L2);
/* One full iteration including the correction (no termination test), for
 * benchmarking; *L2 = norm observed after this cycle's smoothing. */
int mg_vcycle(mg_handle *h, double *L2);
/* `count` iterations back to back with no host synchronisation in between;
 * *L2 = norm observed in the last one. */
int mg_vcycles(mg_handle *h, unsigned count, double *L2);
/* GaussSeidelSolver::solve() (cpp/PotentialSolver.cpp:235-291) on the fine grid
 * with the red-black ordering: one SOR sweep per iteration; in iterations with
 * it % 25 == 0 the norm sqrt(sum(R^2)/(ni*nj*nk)) of the residual over the
 * INTERIOR nodes is tested against tol.  *iterations = sweeps performed, *L2 =
 * last norm evaluated.  Works on any handle (depth 0 allocates no coarse grids). */
int mg_gs_solve(mg_handle *h, unsigned max_it, double tol, int *converged,
                unsigned *iterations, double *L2);

/* PotentialSolver::computeEF() (cpp/PotentialSolver.cpp:418-458) on the
 * device-resident phi: E = -grad(phi), central differences inside, 3-point
 * one-sided differences on the faces.  The field stays on the device until
 * mg_get_ef copies it out as [i][j][k][3] doubles (the layout of World::ef,
 * Field_<double3>); it is invalidated by mg_set_phi and by the solve calls. */
int mg_compute_ef(mg_handle *h);
int mg_get_ef(mg_handle *h, double *ef_flat);

/* Page-locked host memory for the arrays that cross the boundary.  mg_set_rho / mg_set_phi /
 * mg_get_phi / mg_get_ef accept any host pointer; from ordinary (pageable) memory the copy is
 * staged by the runtime at about a third of the host link's rate.  The reference keeps its
 * fields for the whole run (World::phi, World::rho, cpp/World.h:92-96), so an integrator locks
 * them once: mg_host_register pins caller-owned memory in place (undone by mg_host_unregister
 * before it is freed), mg_host_alloc / mg_host_free hand out page-locked blocks. */
int mg_host_register(void *ptr, uint64_t bytes);
int mg_host_unregister(void *ptr);
int mg_host_alloc(void **out, uint64_t bytes);
int mg_host_free(void *ptr);

/* Zero R_l and eps_l: the state of a freshly constructed reference solver
 * (cpp/Field.h:72).  The reference never does this after construction. */
int mg_reset_state(mg_handle *h);

/* Copy a level's work array to the host, for state comparison with the
 * reference members.  which: 0 = R_l (dims n>>l; R_0 only with
 * MG_FLAG_KEEP_FINE_RESIDUAL), 1 = eps_l (dims (n>>l)+1 for l >= 1; eps_0 does not
 * exist on the device, the fine prolongation is fused with the correction).
 * dims_out receives the array dims.  Note: R_1 is refreshed together with the
 * residual norm, i.e. also in the iteration that converges (the reference
 * restricts only after the termination test); no later computation reads the
 * difference. */
int mg_get_level_array(mg_handle *h, int which, int level, double *out_flat,
                       int64_t dims_out[3]);

int mg_set_profiling(mg_handle *h, int on);
int mg_get_timers(mg_handle *h, mg_timers *out);
int mg_reset_timers(mg_handle *h);
/* Device time of the last mg_vcycles()/mg_vcycle() call in milliseconds
 * (HIP events on the handle's stream around the whole call). */
int mg_last_call_ms(mg_handle *h, double *ms);

/* ---- single operators on caller-supplied flat host arrays ---------------
 * Test entry points: each uploads its inputs, runs the same kernel the solver
 * uses, downloads the result.  `device` is the HIP ordinal. */

/* R = A*phi - rho/EPS_0 inside, R = phi on the six faces; *sumsq = sum R^2
 * (cpp/PotentialSolver.cpp:3792-3816). */
int mg_op_residual(int device, const int64_t n[3], const double dh[3],
                   const double *phi, const double *rho, double *R_out,
                   double *sumsq);
/* The fused form the solver uses by default: R_coarse = restrict27(residual)
 * and *sumsq in one pass (cpp/PotentialSolver.cpp:3792-3839).  R_fine_out may be
 * NULL (the fine residual is then never stored); R_coarse (dims n>>1) is
 * read-modify-written like in mg_op_restrict27. */
int mg_op_residual_restrict(int device, const int64_t n[3], const double dh[3],
                            const double *phi, const double *rho, double *R_fine_out,
                            double *R_coarse, double *sumsq);
/* 27-point full weighting, injection where an index is 0
 * (cpp/PotentialSolver.cpp:3822-3839).  R_coarse has dims n_fine>>1 and is
 * read-modify-written: entries the loop does not reach keep their values. */
int mg_op_restrict27(int device, const int64_t n_fine[3], const double *R_fine,
                     double *R_coarse);
/* The "B" classes' 7-point restriction (cpp/PotentialSolver.cpp:704-715); same array
 * conventions as mg_op_restrict27. */
int mg_op_restrict7(int device, const int64_t n_fine[3], const double *R_fine,
                    double *R_coarse);
/* Trilinear prolongation, overwriting eps_fine on 0..n-2 per axis
 * (cpp/PotentialSolver.cpp:4170-4206).  n = size of the fine level (loop
 * bound), fine_dims = dims of eps_fine (n for the finest level, n+1 for a
 * coarse level), coarse dims = (n>>1)+1. */
int mg_op_prolong(int device, const int64_t n[3], const int64_t fine_dims[3],
                  const double *eps_coarse, double *eps_fine);
/* phi -= eps (cpp/PotentialSolver.cpp:4209-4214). */
int mg_op_correct(int device, int64_t count, double *phi, const double *eps);
/* Fused fine prolongation + correction: phi -= P(eps_coarse). */
int mg_op_prolong_correct(int device, const int64_t n[3], const double *eps_coarse,
                          double *phi);
/* Red-black GS-SOR sweeps on u (dims u_dims) with right-hand side rhs (dims
 * rhs_dims) over 1..n-2 per axis; c = {cx,cy,cz}; rhs_is_rho != 0 divides rhs by
 * EPS_0 first (the fine-level form, cpp/PotentialSolver.cpp:3780-3786; coarse
 * form :3929-3935). */
int mg_op_smooth_rb(int device, const int64_t n[3], const int64_t u_dims[3],
                    const int64_t rhs_dims[3], const double c[3], double omega,
                    int sweeps, int rhs_is_rho, const double *rhs, double *u);
/* E = -grad(phi) (cpp/PotentialSolver.cpp:418-458); ef receives
 * n[0]*n[1]*n[2]*3 doubles, [i][j][k][component]. */
int mg_op_compute_ef(int device, const int64_t n[3], const double dh[3],
                     const double *phi, double *ef);
/* sum of x[p]^2 (cpp/PotentialSolver.cpp:3810-3816) */
int mg_op_sumsq(int device, int64_t count, const double *x, double *out);

/* ---- device-level operators for a slab-partitioned solve ------------------
 * One process per GPU owns a slab of i-planes of every distributed level plus
 * halo planes; the host orchestrates (multigrid_amd/dist.py) and moves halo
 * planes between ranks with RCCL send/recv.  These entry points run ONE
 * operator on arrays the caller owns in device memory, asynchronously on the
 * caller's HIP stream.  They have no counterpart in the reference, which is a
 * single-threaded CPU program; each applies the same loop nest as the solver
 * (see the mg_op_* twin for the reference lines) to a range of global planes.
 *
 * mg_dev_array describes planes [i0, i0+planes) of a level's global array:
 * fp64, [plane][nj rows][pitch], pitch = nk rounded up to even, 16-byte
 * aligned base.  n[3] is always the GLOBAL size of the level the loop bounds
 * refer to; lo/hi are global plane indices [lo, hi) the call produces. */
typedef struct mg_dev_array {
    void *ptr;
    int64_t planes, nj, nk;
    int64_t i0;
} mg_dev_array;

/* Row pitch in doubles for rows of nk nodes. */
int64_t mg_dev_pitch(int64_t nk);
/* {cx, cy, cz, 2cx+2cy+2cz} of level `level` (cpp/PotentialSolver.cpp:3728-3750); host only. */
int mg_level_coef(const double dh[3], int level, double out[4]);
/* v /= EPS_0 over the whole array. */
int mg_dev_div_eps0(void *stream, const mg_dev_array *v);
/* 1 when mg_dev_smooth_rb on an array of this shape runs the out-of-place single-pass kernel
 * (result in alt): one sweep may then be issued as several disjoint plane ranges, e.g. the
 * planes next to a neighbour first so that their halo exchange overlaps the rest.  The ptr
 * member is not read. */
int mg_dev_sweep_out_of_place(const mg_dev_array *u);
/* `sweeps` red-black sweeps of planes [lo,hi) of u (right-hand side rhs, level size n).
 * The sweep may be out of place: alt has u's shape; *result_in_alt = 1 when the result
 * landed in alt (the caller swaps its references).  Out-of-place sweeps read planes
 * [lo-2, hi+2) of u as far as they exist, so with halos two planes wide one sweep needs
 * one halo exchange.  sweeps > 1 runs the sweeps one after the other, all on [lo,hi) -- for arrays
 * held whole on one GPU -- except that two sweeps go into ONE launch where
 * mg_dev_sweeps_in_pairs says so (it then reads planes [lo-4, hi+4) of u).
 * [lo2,hi2) (empty when lo2 == hi2) is a second range, as long as the first and above it, swept
 * in the same launch -- the planes next to the other slab neighbour; it needs sweeps == 1 and
 * the out-of-place kernel (mg_dev_sweep_out_of_place). */
int mg_dev_smooth_rb(void *stream, const mg_dev_array *u, const mg_dev_array *alt,
                     const mg_dev_array *rhs, const int64_t n[3], const double c[4], double omega,
                     int sweeps, int64_t lo, int64_t hi, int64_t lo2, int64_t hi2, int *result_in_alt);
/* Coarse planes [lo/2, (hi+1)/2) of Rc = restrict27(residual(phi, b)) for fine planes
 * [lo,hi) (lo even; hi even unless it is the last plane of the level), per-workgroup partial sums of the squared fine
 * residual into partials (capacity `capacity` doubles); *count = partials written.
 * Reads phi planes [lo-2, hi+1). */
int mg_dev_residual_restrict(void *stream, const mg_dev_array *phi, const mg_dev_array *b,
                             const mg_dev_array *Rc, const int64_t n[3], const double c[4],
                             double *partials, int64_t capacity, int64_t lo, int64_t hi,
                             int64_t *count);
/* mg_dev_residual_restrict + mg_dev_sum_partials for callers that cannot allocate device memory
 * (code under HIP graph capture, tight loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The per-workgroup partial sums go to a workspace built into the
 * library (a 1 MiB array in each device's global memory, held until the process ends) and
 * sumsq[0] (device memory, or mapped host memory) receives their sum in the same fixed order, so
 * Rc and the sum are bit-identical to the two-call form.  The call allocates and frees nothing,
 * with one exception: the FIRST call on a device looks the workspace's address up, and if no kernel
 * of this library has run on that device before, the runtime loads the library's code there at
 * that moment (which allocates its global data, this workspace included, and cannot be captured).
 * Make one warm-up call per device, outside any capture, before relying on this.  The workspace
 * used is that of the device `stream` belongs to (of the current device for the null stream).
 * A plane range whose launch has more than 131072 workgroups is rejected with MG_ERR_INVALID
 * (2049 x 2049 planes in chunks of up to 64 stay below that).  All calls on one device share the
 * workspace: calls that overlap in time on one device (different streams or threads) race. */
int mg_dev_residual_restrict_noalloc(void *stream, const mg_dev_array *phi, const mg_dev_array *b,
                                     const mg_dev_array *Rc, const int64_t n[3], const double c[4],
                                     int64_t lo, int64_t hi, double *sumsq);
/* out[0] = sum(partials[0..count)), fixed order. */
int mg_dev_sum_partials(void *stream, const double *partials, int64_t count, double *out);
/* Coarse planes [lo,hi) of Rc = restrict27(Rf); nf = size of the fine level. Reads Rf planes [2lo-1, 2hi). */
int mg_dev_restrict27(void *stream, const mg_dev_array *Rf, const mg_dev_array *Rc,
                      const int64_t nf[3], int64_t lo, int64_t hi);
/* Fine planes [lo,hi): out = P(ec) (correct == 0, out = eps of the finer level) or
 * out -= P(ec) (correct != 0, out = phi); n = size of the fine level.  Reads ec planes [lo/2, (hi-1)/2 + 2). */
int mg_dev_prolong(void *stream, const mg_dev_array *ec, const mg_dev_array *out,
                   const int64_t n[3], int64_t lo, int64_t hi, int correct);

/* mg_dev_smooth_rb preceded by the prolongation from ec (the next coarser level's eps):
 * correct != 0: u := u - P(ec) (u = phi, cpp/PotentialSolver.cpp:4170-4214), else u := P(ec)
 * over 0..n-2 (cpp/PotentialSolver.cpp:3940-3976).  The operation is applied to planes
 * [lo-2, hi+2) of u as far as they are stored -- folded into the first sweep's loads where
 * the single-pass kernel runs (halo planes of u are then left as they were: the caller
 * exchanges halos after a sweep anyway), as separate launches otherwise.  ec must hold the
 * coarse planes those fine planes interpolate from.  sweeps >= 1.  [lo2,hi2): second range as
 * in mg_dev_smooth_rb. */
int mg_dev_prolong_smooth_rb(void *stream, const mg_dev_array *ec, int correct, const mg_dev_array *u,
                             const mg_dev_array *alt, const mg_dev_array *rhs, const int64_t n[3],
                             const double c[4], double omega, int sweeps, int64_t lo, int64_t hi,
                             int64_t lo2, int64_t hi2, int *result_in_alt);
/* 1 when mg_dev_prolong_smooth_rb folds the operation into the first sweep's loads for an array
 * of this shape (u itself is then left untouched, and a level may be swept range by range with
 * overlapping read neighbourhoods); 0 when it applies the operation to u in place first.  The
 * ptr member is not read. */
int mg_dev_onload_folds(const mg_dev_array *u, int correct);
/* 1 when mg_dev_smooth_rb with sweeps == 2 on planes [lo,hi) of arrays of these shapes does both
 * sweeps in ONE launch, reading planes [lo-4, hi+4) of u and [lo-3, hi+3) of rhs as far as the
 * level has them (they must be stored); otherwise the two sweeps run one after the other, both on
 * [lo,hi).  A caller that narrows its plane range from sweep to sweep may hand over two sweeps at
 * once only when this holds.  The ptr members are not read.  No counterpart in the reference. */
int mg_dev_sweeps_in_pairs(const mg_dev_array *u, const mg_dev_array *rhs, const int64_t n[3], int64_t lo, int64_t hi);
/* Planes [lo,hi) of E = -grad(phi) (cpp/PotentialSolver.cpp:418-458).  ef is a dense device
 * array [plane][nj][nk][3] whose plane 0 is global plane ef_i0 (no row padding).  Reads phi
 * planes [lo-1, hi+1), plus plane 2 when lo == 0 and plane n-3 when hi == n. */
int mg_dev_compute_ef(void *stream, const mg_dev_array *phi, void *ef, int64_t ef_i0,
                      const int64_t n[3], const double dh[3], int64_t lo, int64_t hi);

/* ---- slab-partitioned solve, one rank per GPU --------------------------------
 * The cycle of mg_solve on a grid whose i-planes are divided among `world` ranks, each bound to
 * one GPU and driven by one host thread or process.  No counterpart in the reference (one CPU
 * thread, cpp/PotentialSolver.cpp:3756-4216).  phi (and E) after any number of cycles is bit-identical
 * to the undivided handle's.  The residual norm is not: each rank sums the squares of its slab and
 * the sums are all-reduced, which associates the additions differently from the one-GPU sum, so
 * L2 agrees to rounding (~1e-15 relative) and a solve whose norm passes within that distance of
 * tol may stop one cycle earlier or later than the undivided handle.
 * Every rank of a group makes the same sequence of mg_dist_* calls (create, vcycles / solve,
 * destroy are collective).  Halo planes travel by RCCL send/recv (MG_TRANSPORT_RCCL: rank 0 makes
 * a unique id with mg_dist_unique_id, the caller hands the same 128 bytes to every rank -- over
 * MPI, a TCP store, torch.distributed, or plain memory between threads), or between ranks that are
 * threads of one process by device-to-device copies (MG_TRANSPORT_LOCAL with a shared
 * mg_local_group; ranks may share a GPU). */
#define MG_UNIQUE_ID_BYTES 128
#define MG_TRANSPORT_RCCL 0
#define MG_TRANSPORT_LOCAL 1
/* No transfers at all: ONE rank of a world of `world` ranks, alone, running exactly the launches
 * and stream dependencies it would run there (halo planes keep what they hold, sums stay local:
 * the results mean nothing).  For timing one rank's share of a partitioned solve on one GPU. */
#define MG_TRANSPORT_NONE 2
typedef struct mg_dist mg_dist;
typedef struct mg_local_group mg_local_group;
typedef struct mg_dist_config {
    mg_config base;      /* the GLOBAL grid (n, dh, depth, nu, omega); device = this rank's GPU; flags 0 or MG_FLAG_FAST_SMOOTHER */
    int32_t rank, world;
    int32_t coarse_cut;  /* levels >= this are held and solved whole by every rank; 0 = first level with <= 128 planes */
    int32_t transport;   /* MG_TRANSPORT_* */
    int32_t no_overlap;  /* bit 0: exchange halos after whole sweeps instead of under their interior part;
                            bit 1: exchange after every sweep on every level (no deep halos, see mg_dist_deep_halos) */
    uint8_t unique_id[MG_UNIQUE_ID_BYTES];  /* MG_TRANSPORT_RCCL */
    mg_local_group *local_group;            /* MG_TRANSPORT_LOCAL */
} mg_dist_config;

int mg_dist_unique_id(uint8_t out[MG_UNIQUE_ID_BYTES]);
/* hold_back != 0: every halo exchange completes only when its planes are next needed (the latest
 * moment the schedule allows) instead of when it is started -- for tests of the schedule. */
int mg_local_group_create(int world, int hold_back, mg_local_group **out);
int mg_local_group_destroy(mg_local_group *g);
int mg_dist_create(const mg_dist_config *cfg, mg_dist **out);
int mg_dist_destroy(mg_dist *d);
/* The rank's share: *cut, the fine planes it owns [own[0], own[1]), and the planes of phi and of
 * rho it stores and must be given (owned + halo planes). */
int mg_dist_plan(mg_dist *d, int32_t *cut, int64_t own[2], int64_t stored_phi[2], int64_t stored_rho[2]);
/* *deep = 1 when the levels 1..cut-1 are smoothed on deep halos: each rank receives once per cycle
 * the planes of R_l that all sweeps of level l reach beyond its slab (two per sweep and side) and
 * computes eps_l on them as its neighbour does, instead of exchanging two planes after every
 * sweep.  Chosen at creation when every rank finds those planes on its two neighbours and bit 1
 * of mg_dist_config.no_overlap is clear; the results are the same bits either way.  No counterpart
 * in the reference. */
int mg_dist_deep_halos(mg_dist *d, int32_t *deep);
/* The same decision and the plane counts behind it from the configuration alone (base.n, base.depth,
 * base.nu, rank, world, coarse_cut, no_overlap are read; no device, communicator or group is
 * touched): *deep and *cut as a rank created from `cfg` reports them; own = the planes of R_level
 * the rank owns; on deep halos and for level in 1..cut-1, first / last = the planes of eps_level the
 * rank forms before the level's first sweep and holds after its last, and halo = the planes of
 * R_level {received below own, sent from own's low end, sent from own's high end, received above
 * own} in the one exchange per cycle (zeros otherwise).  What a rank sends one way is what its
 * neighbour receives: a caller can check the whole group's message sizes without a GPU.  Any
 * output pointer may be null.  No counterpart in the reference. */
int mg_dist_exchange_plan(const mg_dist_config *cfg, int32_t level, int32_t *deep, int32_t *cut, int64_t own[2],
                          int64_t first[2], int64_t last[2], int32_t halo[4]);
/* Dense host planes [lo,hi) of the global arrays, rows of nk doubles; any sub-range of the stored
 * planes, so a rank never needs its whole slab in host memory.  rho is stored as rho/EPS_0. */
int mg_dist_set_rho_planes(mg_dist *d, int64_t lo, int64_t hi, const double *dense);
int mg_dist_set_phi_planes(mg_dist *d, int64_t lo, int64_t hi, const double *dense);
/* Planes [lo,hi) within the owned range. */
int mg_dist_get_phi_planes(mg_dist *d, int64_t lo, int64_t hi, double *dense);
/* As mg_vcycles / mg_solve; every rank receives the same norm and the same stopping decision. */
int mg_dist_vcycles(mg_dist *d, unsigned count, double *L2);
int mg_dist_solve(mg_dist *d, unsigned max_it, double tol, int *converged, unsigned *cycles, double *L2);
/* E = -grad(phi) on the rank's owned planes (collective: phi's halo planes are exchanged first), as
 * PotentialSolver::computeEF (cpp/PotentialSolver.cpp:418-458); then planes [lo,hi) within the
 * owned range as dense [plane][nj][nk][3]. */
int mg_dist_compute_ef(mg_dist *d);
int mg_dist_get_ef_planes(mg_dist *d, int64_t lo, int64_t hi, double *dense);
int mg_dist_sync(mg_dist *d);
/* Gives up this rank's communication (ncclCommAbort; the in-process group is released): pending
 * transfers end, host calls blocked on them return, and every later collective call on the rank
 * returns MG_ERR_COMM.  May be called from any thread, also while the rank's own thread is inside
 * a call -- it is how a supervisor releases the ranks of a group one of which has failed.  A rank
 * whose own collective call fails does this to itself.  Only mg_dist_destroy is useful afterwards. */
int mg_dist_abort(mg_dist *d);
/* MG_TRANSPORT_NONE only (other ranks report *count = 0): the record of the communication calls the
 * rank would have made over RCCL since it was created, built from the same message lists the RCCL
 * transport issues.  An entry is five numbers {what, peer, array, lo, hi}: what = 0 a halo exchange
 * begins (lo, hi = the owned planes it is around), 4 an all-gather begins (lo, hi = the rank's share),
 * 1 / 2 a send to / a receive from `peer` of the global planes [lo, hi) within that call, 3 an
 * all-reduce of one double; array = 16 * level + {0 phi, 1 its out-of-place twin, 2 rho, 3 R, 4 eps,
 * 5 its twin, 6 / 7 / 8 the whole-level R, eps and twin}.  Up to `capacity` entries are copied,
 * *count is the number recorded.  With the records of all ranks of a world (taken one rank at a time,
 * on one GPU) a caller can check that every send meets its receive: same call, same planes, same
 * array.  No counterpart in the reference. */
int mg_dist_comm_log(mg_dist *d, int64_t *entries, int64_t capacity, int64_t *count);
/* Device time of the last vcycles/solve call and bytes this rank has sent so far. */
int mg_dist_stats(mg_dist *d, double *last_ms, uint64_t *bytes_sent);

#ifdef __cplusplus
}
#endif
#endif /* MG_HIP_H */
