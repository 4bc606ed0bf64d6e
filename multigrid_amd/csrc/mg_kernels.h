// Internal launcher interface between the C ABI (mg_api.hip) and the gfx950
// kernels (mg_kernels.hip, mg_sweep.hip, mg_resrest.hip).  Not part of the public
// boundary.
//
// Index space: kernels think in GLOBAL node indices of a level.  A View may be a
// slab of the global array (planes [i0, i0+d[0]) of its first axis, e.g. the
// planes one GPU owns plus halo planes); at() takes the global i.  Every launcher
// takes the global plane range [lo, hi) it is to produce, so the same kernels
// serve the single-GPU case (i0 = 0, range = whole array) and a slab partition.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgk {

constexpr double kEps0 = 8.85418782e-12;  // reference cpp/World.h:14

// A 3-D fp64 array (or a slab of one) in device memory.
struct View {
    double *p;
    int64_t d[3];    // stored planes, rows, columns
    int64_t si, sj;  // strides in doubles; sj >= d[2] is the (even) row pitch
    int64_t i0;      // global index of stored plane 0
    __host__ __device__ int64_t at(int64_t i, int64_t j, int64_t k) const { return (i - i0) * si + j * sj + k; }
    __host__ __device__ int64_t i_end() const { return i0 + d[0]; }
    int64_t elems() const { return d[0] * si; }  // allocated doubles
};

struct IRange { int64_t lo, hi; };  // global planes [lo, hi)

// Number of i-chunks for a marching kernel: `tiles` (j,k) windows, `planes` planes to produce,
// `extra` planes each chunk processes beyond its own (halo planes at the chunk ends).  All
// workgroups of a cohort start together and march at the same pace, so the run takes about
// rounds x planes-per-chunk; pick the chunk count that minimises it for `slots` workgroups
// resident at once (2 per CU).  CI is rounded up to a multiple of `align`.  The chunk index is
// the fastest-varying part of the workgroup index, so with a multiple of 8 chunks all windows
// of a chunk land on one XCD and share its L2 (measured: 11 chunks instead of 8 cost the 513^3
// sweep 13 %); multiples of 8 are preferred unless another count is clearly cheaper.
inline int pick_chunks(int64_t tiles, int64_t planes, int extra, int align = 1, int64_t slots = 512,
                       bool prefer8 = true)
{
    int best = 1, best8 = 0;
    double cost_best = 1e300, cost8 = 1e300;
    for (int c = 1; c <= 64 && (planes + c - 1) / c >= 4; c++) {
        int64_t ci = (planes + c - 1) / c;
        ci = (ci + align - 1) / align * align;
        const int64_t rounds = (tiles * c + slots - 1) / slots;
        const double cost = (double)rounds * (double)(ci + extra);
        if (cost < cost_best - 1e-9) { cost_best = cost; best = c; }
        if (c % 8 == 0 && cost < cost8 - 1e-9) { cost8 = cost; best8 = c; }
    }
    // a multiple of 8 keeps each chunk on one XCD; give that up only for a clear gain
    return (prefer8 && best8 && cost8 <= 1.08 * cost_best) ? best8 : best;
}

// Workgroup -> (chunk, window) assignment of the marching kernels.  Workgroups are dealt to the
// 8 XCDs round-robin by index (each XCD has its own L2), and neighbouring windows read the same
// halo lines: `blocked` hands every XCD one contiguous run of the (chunk, window-row, window-column)
// order, so a window's neighbours sit on the same L2 whatever the chunk count.  The grid is
// padded to 8 x ceil(total/8) workgroups; item() returns -1 for the padding.
struct WgOrder {
    int total;    // chunks x windows
    int blocked;  // 1: XCD-blocked order; 0: plain index (chunk fastest)
    __host__ __device__ unsigned grid() const { return blocked ? 8u * (unsigned)((total + 7) / 8) : (unsigned)total; }
#if defined(__HIPCC__)
    __device__ int item(unsigned b) const
    {
        if (!blocked) return (int)b;
        const int per = (total + 7) / 8;
        const int o = (int)(b % 8u) * per + (int)(b / 8u);
        return o < total ? o : -1;
    }
#endif
};
bool wg_order_blocked();

// Row pitch used for every device array: rows start 16-byte aligned.
int64_t pitch_for(int64_t nk);
int exp_knob(const char *name, int dflt);  // experiment switches from the environment (read once each)

// Stencil coefficients of a level: 1/h^2 per axis, d = 2cx+2cy+2cz, and what div_d() needs.
struct Coef {
    double cx, cy, cz, d;
    double rd;     // 1.0/d, correctly rounded
    int fma_div;   // 1: div_d() may use the reciprocal form (see div_d)
    // The smoother's update with the division and omega folded into the coefficients
    // (MG_FLAG_FAST_SMOOTHER; see sor_n): fb = omega*rd, fx = (omega*cx)*rd, fy, fz alike, fu = 1 - omega.
    int fast;
    double fb, fx, fy, fz, fu;
};
// Fills rd / fma_div from d.
Coef make_coef(double cx, double cy, double cz, double d);
// c with the folded coefficients of sor_n<true> for this omega, and fast = 1.
Coef with_fast_smoother(Coef c, double omega);
// on: coefficients made from now on divide with the division instruction (fma_div = 0).
void set_exact_division(bool on);

#if defined(__HIPCC__)
// x / c.d, correctly rounded (bit-identical to the IEEE division the reference performs,
// cpp/PotentialSolver.cpp:3783), from the correctly rounded reciprocal rd instead of the
// ~15-instruction division sequence.  q1 = fma(x - q0*d, rd, q0) is within half an ulp (plus
// second-order terms) of x/d, hence a faithful rounding; its residual r1 = x - q1*d is then exact
// in one FMA, and Markstein's theorem (the reciprocal is correctly rounded, the significand of d
// is not all ones -- checked on the host) makes fma(r1, rd, q1) the correctly rounded quotient.
// The argument needs every intermediate to stay in the normal range: make_coef admits
// 2^-100 < d < 2^100 only, and an x outside 2^-800 <= |x| < 2^800 (zeros, whose sign the FMA
// chain would lose, infinities and NaN included) takes the division instruction; the range test
// is one unsigned compare on the exponent bits.  Explicit fma() calls are not contractions:
// -ffp-contract=off does not affect them.
__device__ __forceinline__ double div_d(double x, const Coef &c)
{
    constexpr uint32_t kLo = (1023u - 800u) << 20, kSpan = 1600u << 20;
    const uint32_t mag = (uint32_t)__double2hiint(x) & 0x7fffffffu;
    if (!c.fma_div || mag - kLo >= kSpan) return x / c.d;
    const double q0 = x * c.rd;
    const double q1 = fma(fma(-q0, c.d, x), c.rd, q0);
    return fma(fma(-q1, c.d, x), c.rd, q1);
}

// div_d() for N numerators at once: the reciprocal form runs unconditionally on all of them
// (straight-line code the N chains interleave in) and the rare numerator outside its range is
// redone with the division instruction behind ONE branch, instead of a branch per quotient.
template <int N>
__device__ __forceinline__ void div_d_n(const double (&x)[N], const Coef &c, double (&g)[N])
{
    constexpr uint32_t kLo = (1023u - 800u) << 20, kSpan = 1600u << 20;
    bool redo = !c.fma_div;
#pragma unroll
    for (int n = 0; n < N; n++) {
        const double q0 = x[n] * c.rd;
        const double q1 = fma(fma(-q0, c.d, x[n]), c.rd, q0);
        g[n] = fma(fma(-q1, c.d, x[n]), c.rd, q1);
        redo |= ((uint32_t)__double2hiint(x[n]) & 0x7fffffffu) - kLo >= kSpan;
    }
    if (redo) {
#pragma unroll
        for (int n = 0; n < N; n++)
            if (!c.fma_div || ((uint32_t)__double2hiint(x[n]) & 0x7fffffffu) - kLo >= kSpan) g[n] = x[n] / c.d;
    }
}

// The SOR update of N points: b = right-hand side, si / sj / sk = u[i-1]+u[i+1], u[j-1]+u[j+1],
// u[k-1]+u[k+1] (each summed in that order), u0 = the current value.
//   FAST = false: the reference's expression in its order (cpp/PotentialSolver.cpp:3780-3786, :3929-3935),
//       g = (((b + cx*si) + cy*sj) + cz*sk) / d;  u = u0 + omega*(g - u0)
//     bit-identical to the scalar CPU evaluation (17 operations and a correctly rounded quotient);
//   FAST = true: the same update as one chain of fused multiply-adds on coefficients that carry
//     the division and omega,
//       u = fma(fz, sk, fma(fy, sj, fma(fx, si, fma(fb, b, fu*u0))))
//     8 operations; equal to the other form up to rounding (a few ulp per update), which the
//     smoother of an iteration that measures its residual exactly can afford.
template <bool FAST, int N>
__device__ __forceinline__ void sor_n(const double (&b)[N], const double (&si)[N], const double (&sj)[N],
                                      const double (&sk)[N], const double (&u0)[N], const Coef &c, double omega,
                                      double (&un)[N])
{
    if constexpr (FAST) {
#pragma unroll
        for (int n = 0; n < N; n++)
            un[n] = fma(c.fz, sk[n], fma(c.fy, sj[n], fma(c.fx, si[n], fma(c.fb, b[n], c.fu * u0[n]))));
    } else {
        double x[N], g[N];
#pragma unroll
        for (int n = 0; n < N; n++) x[n] = b[n] + c.cx * si[n] + c.cy * sj[n] + c.cz * sk[n];
        div_d_n<N>(x, c, g);
#pragma unroll
        for (int n = 0; n < N; n++) un[n] = u0[n] + omega * (g[n] - u0[n]);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a workgroup of up to 1024 threads; result valid in thread 0.
// Fixed order: lanes by shuffle tree, then waves 0..nw-1 left to right.
__device__ __forceinline__ double block_sum(double v, double *lds)
{
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    v = wave_sum(v);
    if ((tid & 63) == 0) lds[tid >> 6] = v;
    __syncthreads();
    double t = 0;
    if (tid == 0)
        for (int w = 0; w < nw; w++) t += lds[w];
    return t;
}
#endif

// v = v / EPS_0 in place over the whole view: turns an uploaded rho into the rhs b.
void div_eps0(hipStream_t s, View v);

// v = dense (rows of d[2] doubles, device memory), divided by EPS_0 when div; dense = v.
void repitch_in(hipStream_t s, View v, const double *dense, bool div);
void repitch_out(hipStream_t s, double *dense, View v);

// One colour of red-black GS-SOR on u, updating planes of r within 1..n-2 (n = level size).
void smooth_colour(hipStream_t s, View u, View rhs, const int64_t n[3], Coef c, double omega,
                   int colour, IRange r);
// One full red-black sweep (both colours) in a single pass, out of place: planes r of
// u_out = sweep(u_in); points outside 1..n-2 are copied through.  u_in must hold planes
// [r.lo-2, r.hi+2) as far as they exist in the global array (gI planes).
// A non-empty r2 (same length as r, disjoint, above it) is swept in the same launch.
void sweep_rb_fused(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c,
                    double omega, IRange r, IRange r2 = IRange{0, 0});
// Two full sweeps in one pass (half the traffic per sweep): planes r of u_out = sweep(sweep(u_in)).
// u_in must hold planes [r.lo-4, r.hi+4) and rhs planes [r.lo-3, r.hi+3) as far as they exist.
void sweep_rb_fused2(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c,
                     double omega, IRange r);
// The same with the operation that precedes the sweep applied to u_in as it is loaded:
// correct: u_in := u_in - P(ec) (fine grid, phi -= eps_h); otherwise u_in := P(ec) over
// 0..n-2 (a coarse level's prolongated initial guess).  ec = eps of the next coarser level.
// outside_zero (prolongation only): the caller guarantees that u_in is zero outside 0..n-2 per
// axis (true of every eps_l the solver owns); u_in is then not read at all.
void sweep_rb_fused_onload(hipStream_t s, View uin, View uout, View rhs, View ec, bool correct,
                           const int64_t n[3], Coef c, double omega, IRange r, IRange r2 = IRange{0, 0},
                           bool outside_zero = false);
// The same by 3-D tiles instead of a march along i (mg_tile.hip): for levels with short rows and
// for plane ranges too short to march along.  Same contract as sweep_rb_fused.
void sweep_rb_tile(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega,
                   IRange r, IRange r2 = IRange{0, 0});
// Two sweeps of planes r in one launch by tiles; u_in must hold planes [r.lo-4, r.hi+4) as far as the
// level has them, rhs planes [r.lo-3, r.hi+3).
void sweep_rb_tile2(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega, IRange r);
// True when smooth_rb on an array of this shape (with a twin to write to) runs an out-of-place
// single-pass kernel: disjoint plane ranges of one sweep are then independent launches.
bool sweep_is_out_of_place(const View &u);
// True when prolong_smooth_rb folds the prolongation (correct: the correction) into the loads of
// the first sweep and leaves u itself untouched; otherwise it applies it in place to the planes
// around the range, and a caller that sweeps a level range by range must not let those overlap.
bool onload_folds(const View &u, bool correct);
// `sweeps` red-black sweeps preceded by the prolongation from ec (correct: by phi -= P(ec)):
// fused into the first sweep where the single-pass kernel is used, separate launches otherwise.
// outside_zero: see sweep_rb_fused_onload.
int prolong_smooth_rb(hipStream_t s, View ec, bool correct, View *u, View *alt, View rhs,
                      const int64_t n[3], Coef c, double omega, int sweeps, IRange r, IRange r2 = IRange{0, 0},
                      bool outside_zero = false);
// `sweeps` red-black sweeps on planes r of *u.  May ping-pong between *u and *alt (same
// shape); on return *u is the view holding the result.  Returns the kernels launched.
// r2: second plane range for the same (single, out-of-place) sweep; see sweep_rb_fused.
int smooth_rb(hipStream_t s, View *u, View *alt, View rhs, const int64_t n[3], Coef c, double omega,
              int sweeps, IRange r, IRange r2 = IRange{0, 0});
// Whether smooth_rb does two sweeps on planes r in ONE launch (reading planes r.lo-4 .. r.hi+3 of
// *u): a caller that narrows the range from sweep to sweep may only hand over two sweeps at once
// when this holds -- otherwise they would run one after the other, both on r.
bool sweeps_in_pairs(const View &u, const View &alt, const View &rhs, const int64_t n[3], IRange r);

// R = A*phi - b inside, R = phi on the faces of the n-sized level, for planes r; per-workgroup
// sums of squares go to partials[0..residual_partials(n, r)).
int64_t residual_partials(const int64_t n[3], IRange r);
void residual(hipStream_t s, View phi, View b, View R, const int64_t n[3], Coef c, double *partials,
              IRange r);
// Fused residual + sum of squares + restriction: coarse planes [r.lo/2, ...) of
// Rc = restrict27(residual(phi, b)) for fine planes r (r.lo even); the fine residual is
// written only when Rf.p is non-null.  phi must hold planes [r.lo-2, r.hi+1).
// A non-empty r2 (same length as r, disjoint, above it, r2.lo even) goes through the same launch.
int64_t residual_restrict_partials(const int64_t n[3], IRange r, IRange r2 = IRange{0, 0});
void residual_restrict(hipStream_t s, View phi, View b, View Rf, View Rc, const int64_t n[3], Coef c,
                       double *partials, IRange r, IRange r2 = IRange{0, 0});
// out[0] = sum of partials[0..count) in a fixed order.
void sum_partials(hipStream_t s, const double *partials, int64_t count, double *out);
// partial sums of x^2 over a flat array
int64_t sumsq_partials(int64_t count);
void sumsq(hipStream_t s, const double *x, int64_t count, double *partials);
// Coarse planes rc of Rc = restrict27(Rf); nf = size of the fine level.
void restrict27(hipStream_t s, View Rf, View Rc, const int64_t nf[3], IRange rc);
// The same with the 7-point weights of the reference's "B" solver classes.
void restrict7(hipStream_t s, View Rf, View Rc, const int64_t nf[3], IRange rc);
// trilinear prolongation of ec onto planes r of ef (within 0..n-2 per axis)
void prolong(hipStream_t s, View ec, View ef, const int64_t n[3], IRange r);
// phi -= prolong(ec) on planes r (within 1..n-2 per axis)
void prolong_correct(hipStream_t s, View ec, View phi, const int64_t n[3], IRange r);
void correct(hipStream_t s, double *phi, const double *eps, int64_t count);

// E = -grad(phi) for planes r (reference cpp/PotentialSolver.cpp:418-458).  ef is dense
// [plane][nj][nk][3] and its plane 0 is global plane ef_i0.  phi must hold planes
// [r.lo-1, r.hi+1) as far as they exist, plane 2 when r contains plane 0 and plane n-3 when r
// contains plane n-1.
void compute_ef(hipStream_t s, View phi, double *ef, int64_t ef_i0, const int64_t n[3], const double dh[3],
                IRange r);
// Residual of GaussSeidelSolver's convergence test (reference cpp/PotentialSolver.cpp:270-282):
// per-workgroup partial sums of R^2 over the interior points of planes r,
// R = -phi*d + b + cx*(..) + cy*(..) + cz*(..).
int64_t gs_residual_partials(const int64_t n[3], IRange r);
void gs_residual_sumsq(hipStream_t s, View phi, View b, const int64_t n[3], Coef c, double *partials, IRange r);

}  // namespace mgk
