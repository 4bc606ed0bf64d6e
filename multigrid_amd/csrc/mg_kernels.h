// Internal launcher interface between the C ABI (mg_api.hip) and the gfx950
// kernels (mg_kernels.hip).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgk {

constexpr double kEps0 = 8.85418782e-12;  // reference cpp/World.h:14

// A 3-D fp64 array in device memory: element (i,j,k) at p[i*si + j*sj + k].
// d[] are the logical dims; sj >= d[2] is the (even) row pitch in doubles.
struct View {
    double *p;
    int64_t d[3];
    int64_t si, sj;
    __host__ __device__ int64_t at(int64_t i, int64_t j, int64_t k) const { return i * si + j * sj + k; }
    int64_t elems() const { return d[0] * si; }  // allocated doubles
};

// Row pitch used for every device array: rows start 16-byte aligned.
inline int64_t pitch_for(int64_t nk) { return (nk + 1) & ~int64_t(1); }

struct Coef { double cx, cy, cz, d; };

// Number of partial sums residual()/sumsq() write for a problem of this size.
int64_t residual_partials(const int64_t n[3]);
int64_t sumsq_partials(int64_t count);

// v = v / EPS_0 in place: turns an uploaded rho into the right-hand side b.
void div_eps0(hipStream_t s, View v);

// One colour of red-black GS-SOR on u over 1..n-2 per axis (n = level size).
void smooth_colour(hipStream_t s, View u, View rhs, const int64_t n[3], Coef c,
                   double omega, int colour);
// One full red-black sweep (both colours) in a single pass, out of place:
// u_out = sweep(u_in); elements outside 1..n-2 are copied through.
void sweep_rb_fused(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c,
                    double omega);
// `sweeps` red-black sweeps on *u.  May ping-pong between *u and *alt (same dims);
// on return *u is the view holding the result and *alt the other buffer.
// Returns the number of kernels launched.
int smooth_rb(hipStream_t s, View *u, View *alt, View rhs, const int64_t n[3], Coef c,
               double omega, int sweeps);
// R = A*phi - b inside, R = phi on faces; block partial sums of R^2 -> partials.
void residual(hipStream_t s, View phi, View b, View R, Coef c, double *partials);
// Fused residual + sum of squares + restriction to the next level: Rc =
// restrict27(residual(phi, b)); per-workgroup partial sums of squares of the fine
// residual go to partials[0..residual_restrict_partials(n)).  The fine residual
// itself is written only when Rf.p is non-null.
int64_t residual_restrict_partials(const int64_t n[3]);
void residual_restrict(hipStream_t s, View phi, View b, View Rf, View Rc, Coef c, double *partials);
// out[0] = sum of partials[0..count) in a fixed order.
void sum_partials(hipStream_t s, const double *partials, int64_t count, double *out);
// partial sums of x^2 over a flat array
void sumsq(hipStream_t s, const double *x, int64_t count, double *partials);
void restrict27(hipStream_t s, View Rf, View Rc);
// trilinear prolongation of ec onto ef over 0..n-2 per axis
void prolong(hipStream_t s, View ec, View ef, const int64_t n[3]);
// phi -= prolong(ec) over 1..n-2 per axis
void prolong_correct(hipStream_t s, View ec, View phi, const int64_t n[3]);
void correct(hipStream_t s, double *phi, const double *eps, int64_t count);

}  // namespace mgk
