// gfx950 kernels of the multigrid V-cycle.  Built with -ffp-contract=off: every
// expression below is evaluated in the association order written, with one
// rounding per operation and true fp64 divisions, so results are bit-identical
// to the scalar CPU evaluation of the same expressions.
//
// Thread mapping used throughout: the 64 lanes of a wavefront run along k (the
// contiguous axis), a 256-thread workgroup covers 4 rows in j, blockIdx.z (or a
// march inside the kernel) covers i.
#include "mg_kernels.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace mgk {

namespace {

constexpr int kBX = 64;  // lanes along k
constexpr int kBY = 4;   // rows along j per workgroup
constexpr int kLdsSmoothMaxBytes = 128 * 1024;  // u + rhs of a level handled by k_smooth_lds

__global__ __launch_bounds__(256) void k_div_eps0(View v)
{
    const int64_t k = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t j = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t i = v.i0 + blockIdx.z;
    if (k >= v.d[2] || j >= v.d[1]) return;
    double *p = v.p + v.at(i, j, k);
    p[0] = p[0] / kEps0;
}

// Dense rows of nk doubles (the caller's array, already in device memory) -> rows of the
// view's pitch, optionally divided by EPS_0 on the way; and back.
__global__ __launch_bounds__(256) void k_repitch_in(View v, const double *__restrict__ dense, int div)
{
    const int64_t k = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t j = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t i = blockIdx.z;
    if (k >= v.d[2] || j >= v.d[1]) return;
    const double x = dense[(i * v.d[1] + j) * v.d[2] + k];
    v.p[i * v.si + j * v.sj + k] = div ? x / kEps0 : x;
}

__global__ __launch_bounds__(256) void k_repitch_out(double *__restrict__ dense, View v)
{
    const int64_t k = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t j = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t i = blockIdx.z;
    if (k >= v.d[2] || j >= v.d[1]) return;
    dense[(i * v.d[1] + j) * v.d[2] + k] = v.p[i * v.si + j * v.sj + k];
}

// One colour of GS-SOR.  u_new = u + omega*(g - u),
// g = (((b + cx*(u[i-1]+u[i+1])) + cy*(u[j-1]+u[j+1])) + cz*(u[k-1]+u[k+1])) / d
// (reference cpp/PotentialSolver.cpp:3780-3786 and :3929-3935).
template <bool FAST>
__global__ __launch_bounds__(256) void k_smooth_colour(View u, View rhs, int64_t nj, int64_t nk,
                                                       Coef c, double omega, int colour, int64_t ilo)
{
    const int64_t i = ilo + (int64_t)blockIdx.z;
    const int64_t j = 1 + (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t m = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t k = 1 + ((i + j + 1 + colour) & 1) + 2 * m;
    if (j > nj - 2 || k > nk - 2) return;
    double *p = u.p + u.at(i, j, k);
    const double b[1] = {rhs.p[rhs.at(i, j, k)]}, si[1] = {p[-u.si] + p[u.si]}, sj[1] = {p[-u.sj] + p[u.sj]},
                 sk[1] = {p[-1] + p[1]}, u0[1] = {p[0]};
    double un[1];
    sor_n<FAST, 1>(b, si, sj, sk, u0, c, omega, un);
    p[0] = un[0];
}

// All sweeps of a small level inside one workgroup: u lives in LDS, the two colours are
// separated by workgroup barriers instead of kernel boundaries.  Used for the coarsest grid,
// whose 50 sweeps would otherwise be 100 launches of a few thousand points each.
//
// LDS layout: the two colours are stored apart, each as [i][j][k>>1] with H = (uK+1)/2 entries
// per row.  The six neighbours of a point all have the other colour, and consecutive lanes
// (consecutive points of one colour in a row) then read consecutive LDS words: no bank
// conflicts, where a dense [i][j][k] copy is read at a stride of two doubles (half the banks).
// Every thread keeps its points (at most P per colour) in registers for the whole launch:
// LDS offset, right-hand side (constant over the sweeps) and current value, so a half-sweep is
// six LDS reads, the update and one LDS write per point.
__host__ __device__ inline size_t smooth_lds_doubles(int64_t uI, int64_t uJ, int64_t uK)
{
    return (size_t)(2 * uI * uJ * ((uK + 1) / 2));
}

template <int P, int NT = 1024, bool FAST = false>
__global__ __launch_bounds__(NT) void k_smooth_lds(View u, View rhs, int nI, int nJ, int nK,
                                                   Coef c, double omega, int sweeps)
{
    extern __shared__ __align__(16) double su[];
    const int uI = (int)u.d[0], uJ = (int)u.d[1], uK = (int)u.d[2];
    const int H = (uK + 1) / 2, HALF = uI * uJ * H;
    const int tid = threadIdx.x;
    auto lds_at = [&](int i, int j, int k) { return ((i + j + k) & 1) * HALF + (i * uJ + j) * H + (k >> 1); };
    for (int p = tid; p < uI * uJ * uK; p += NT) {
        const int k = p % uK, j = (p / uK) % uJ, i = p / (uK * uJ);
        su[lds_at(i, j, k)] = u.p[u.at(i, j, k)];
    }
    const int half = (nK - 1) / 2;  // colour positions per row, at most
    const int rows = (nI - 2) * (nJ - 2);
    int off[2][P];      // LDS offset of the point's (i, j, k>>1) entry in the OTHER colour's half, -1 = none
    int kodd[2][P];     // k & 1: the k-1 / k+1 neighbours sit at entries kodd-1 and kodd of that row
    double b[2][P];     // right-hand side
    double cur[2][P];   // current value (only this thread changes it)
#pragma unroll
    for (int colour = 0; colour < 2; colour++)
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int idx = tid + q * NT;
            off[colour][q] = -1;
            kodd[colour][q] = 0;
            b[colour][q] = 0.0;
            cur[colour][q] = 0.0;
            if (idx < rows * half) {
                const int m = idx % half, t = idx / half;
                const int j = 1 + t % (nJ - 2), i = 1 + t / (nJ - 2);
                const int k = 1 + ((i + j + 1 + colour) & 1) + 2 * m;
                if (k <= nK - 2) {
                    off[colour][q] = (1 - colour) * HALF + (i * uJ + j) * H + (k >> 1);
                    kodd[colour][q] = k & 1;
                    b[colour][q] = rhs.p[rhs.at(i, j, k)];
                    cur[colour][q] = u.p[u.at(i, j, k)];
                }
            }
        }
    __syncthreads();
    const int si = uJ * H, sj = H;
    // How many of a wave's P point slots hold a point in any lane (the slots fill in order of q, so
    // the used ones are a prefix).  A wave skips the slots that are empty in all its lanes: with
    // 1372 points per colour on 1024 threads, ten of the sixteen waves have nothing in slot 1.
    int used[2];
#pragma unroll
    for (int colour = 0; colour < 2; colour++) {
        used[colour] = 0;
#pragma unroll
        for (int q = 0; q < P; q++)
            if (__builtin_amdgcn_readfirstlane(__any(off[colour][q] >= 0))) used[colour] = q + 1;
    }
    for (int s = 0; s < sweeps; s++) {
#pragma unroll
        for (int colour = 0; colour < 2; colour++) {
            // a thread without a point reads around some interior entry and discards the result
            auto gather = [&](int q, double &ti, double &tj, double &tk) {
                const int o = off[colour][q] < 0 ? (1 - colour) * HALF + si + sj + 1 : off[colour][q];
                const double *p = su + o;
                const double *pk = p + kodd[colour][q];
                ti = p[-si] + p[si];
                tj = p[-sj] + p[sj];
                tk = pk[-1] + pk[0];
            };
            if (used[colour] == P) {
                double ti[P], tj[P], tk[P], un[P];
#pragma unroll
                for (int q = 0; q < P; q++) gather(q, ti[q], tj[q], tk[q]);
                sor_n<FAST, P>(b[colour], ti, tj, tk, cur[colour], c, omega, un);  // the quotients of a thread's points behind one branch
#pragma unroll
                for (int q = 0; q < P; q++) cur[colour][q] = un[q];
            } else {
#pragma unroll
                for (int q = 0; q < P - 1; q++)
                    if (q < used[colour]) {
                        double ti[1], tj[1], tk[1], un[1];
                        const double bq[1] = {b[colour][q]}, uq[1] = {cur[colour][q]};
                        gather(q, ti[0], tj[0], tk[0]);
                        sor_n<FAST, 1>(bq, ti, tj, tk, uq, c, omega, un);
                        cur[colour][q] = un[0];
                    }
            }
#pragma unroll
            for (int q = 0; q < P; q++)  // own entry: same (i, j, k>>1), this colour's half
                if (off[colour][q] >= 0) su[off[colour][q] + (colour ? HALF : -HALF)] = cur[colour][q];
            __syncthreads();
        }
    }
    for (int p = tid; p < uI * uJ * uK; p += NT) {
        const int k = p % uK, j = (p / uK) % uJ, i = p / (uK * uJ);
        u.p[u.at(i, j, k)] = su[lds_at(i, j, k)];
    }
}

// Residual (reference cpp/PotentialSolver.cpp:3792-3805) with the workgroup's
// sum of squares written to partials[linear block id].
__global__ __launch_bounds__(256) void k_residual(View phi, View b, View R, int64_t ni, Coef c,
                                                  double *__restrict__ partials, int64_t ilo)
{
    __shared__ double lds[4];
    const int64_t nj = phi.d[1], nk = phi.d[2];
    const int64_t k = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t j = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t i = ilo + blockIdx.z;
    double r = 0;
    if (k < nk && j < nj) {
        const double *p = phi.p + phi.at(i, j, k);
        if (i == 0 || i == ni - 1 || j == 0 || j == nj - 1 || k == 0 || k == nk - 1)
            r = p[0] - 0;
        else
            r = p[0] * c.d - b.p[b.at(i, j, k)] - c.cx * (p[-phi.si] + p[phi.si]) -
                c.cy * (p[-phi.sj] + p[phi.sj]) - c.cz * (p[-1] + p[1]);
        R.p[R.at(i, j, k)] = r;
    }
    const double t = block_sum(r * r, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        partials[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

__global__ __launch_bounds__(1024) void k_sum_partials(const double *__restrict__ partials,
                                                       int64_t count, double *__restrict__ out)
{
    __shared__ double lds[16];
    double v = 0;
    for (int64_t p = threadIdx.x; p < count; p += 1024) v += partials[p];
    const double t = block_sum(v, lds);
    if (threadIdx.x == 0) out[0] = t;
}

__global__ __launch_bounds__(256) void k_sumsq(const double *__restrict__ x, int64_t count,
                                               double *__restrict__ partials)
{
    __shared__ double lds[4];
    const int64_t stride = (int64_t)gridDim.x * 256;
    double v = 0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < count; p += stride) v += x[p] * x[p];
    const double t = block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// 27-point full weighting (reference cpp/PotentialSolver.cpp:3822-3839): terms
// summed left to right, first index fastest, third slowest, then / 64.0.
__global__ __launch_bounds__(256) void k_restrict27(View Rf, View Rc, int64_t Ilo)
{
    const int64_t K = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t J = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t I = Ilo + blockIdx.z;
    if (K >= Rc.d[2] || J >= Rc.d[1]) return;
    const double *f = Rf.p + Rf.at(2 * I, 2 * J, 2 * K);
    double v;
    if (I == 0 || J == 0 || K == 0) {
        v = f[0];
    } else {
        const int64_t si = Rf.si, sj = Rf.sj;
#define F(di, dj, dk) f[(di) * si + (dj) * sj + (dk)]
        v = F(-1, -1, -1) + 2 * F(0, -1, -1) + F(1, -1, -1) +
            2 * F(-1, 0, -1) + 4 * F(0, 0, -1) + 2 * F(1, 0, -1) +
            F(-1, 1, -1) + 2 * F(0, 1, -1) + F(1, 1, -1) +
            2 * F(-1, -1, 0) + 4 * F(0, -1, 0) + 2 * F(1, -1, 0) +
            4 * F(-1, 0, 0) + 8 * F(0, 0, 0) + 4 * F(1, 0, 0) +
            2 * F(-1, 1, 0) + 4 * F(0, 1, 0) + 2 * F(1, 1, 0) +
            F(-1, -1, 1) + 2 * F(0, -1, 1) + F(1, -1, 1) +
            2 * F(-1, 0, 1) + 4 * F(0, 0, 1) + 2 * F(1, 0, 1) +
            F(-1, 1, 1) + 2 * F(0, 1, 1) + F(1, 1, 1);
#undef F
        v = v / 64.0;
    }
    Rc.p[Rc.at(I, J, K)] = v;
}

// 7-point restriction of the "B" solver family (reference cpp/PotentialSolver.cpp:704-715,
// :1991-2002): terms summed left to right, then / 12.0; injection where an index is 0.
__global__ __launch_bounds__(256) void k_restrict7(View Rf, View Rc, int64_t Ilo)
{
    const int64_t K = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t J = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t I = Ilo + blockIdx.z;
    if (K >= Rc.d[2] || J >= Rc.d[1]) return;
    const double *f = Rf.p + Rf.at(2 * I, 2 * J, 2 * K);
    double v;
    if (I == 0 || J == 0 || K == 0) {
        v = f[0];
    } else {
        const int64_t si = Rf.si, sj = Rf.sj;
        v = (f[-si] + 2 * f[0] + f[si] + f[-sj] + 2 * f[0] + f[sj] + f[-1] + 2 * f[0] + f[1]) / 12.0;
    }
    Rc.p[Rc.at(I, J, K)] = v;
}

// Prolongation by coarse cell: a thread takes the eight coarse corner values of
// cell (I,J,K) and produces the eight fine points (2I+di, 2J+dj, 2K+dk), as four
// 16-byte pairs along k.  The eight parity cases of the reference
// (cpp/PotentialSolver.cpp:4176-4205) are written out with their shared leading
// partial sums, in the reference's association order.
// CORRECT: out = out - v (out is phi); otherwise out = v (out is eps of the finer level).
struct alignas(16) dpair { double x, y; };

template <bool CORRECT>
__global__ __launch_bounds__(256) void k_prolong(View ec, View out, int64_t nI, int64_t nJ, int64_t nK,
                                                 int64_t ilo, int64_t ihi)
{
    const int64_t K = (int64_t)blockIdx.x * kBX + threadIdx.x;
    const int64_t J = (int64_t)blockIdx.y * kBY + threadIdx.y;
    const int64_t I = (ilo >> 1) + blockIdx.z;  // cells whose fine planes 2I, 2I+1 meet [ilo, ihi)
    // fine points 0..n-2 per axis are written; cell K covers k = 2K, 2K+1.  A cell past that
    // range writes nothing and its far corners lie outside the coarse array: it must not load.
    if (2 * K > nK - 2 || 2 * J > nJ - 2 || 2 * I > nI - 2) return;
    const double *c = ec.p + ec.at(I, J, K);
    const int64_t si = ec.si, sj = ec.sj;
    const double c000 = c[0], c100 = c[si], c010 = c[sj], c110 = c[si + sj];
    const double c001 = c[1], c101 = c[si + 1], c011 = c[sj + 1], c111 = c[si + sj + 1];
    const double s100 = c000 + c100;          // (1,0,0) and prefix of (1,1,*), (1,0,1)
    const double s010 = c000 + c010;          // (0,1,0) and prefix of (0,1,1)
    const double s110 = (s100 + c010) + c110; // (1,1,0) and prefix of (1,1,1)
    double v[2][2][2];                        // [di][dj][dk]
    v[0][0][0] = c000;
    v[1][0][0] = 0.5 * s100;
    v[0][1][0] = 0.5 * s010;
    v[0][0][1] = 0.5 * (c000 + c001);
    v[1][1][0] = 0.25 * s110;
    v[1][0][1] = 0.25 * ((s100 + c001) + c101);
    v[0][1][1] = 0.25 * ((s010 + c001) + c011);
    v[1][1][1] = 0.125 * ((((s110 + c001) + c101) + c011) + c111);
    const bool k1 = 2 * K + 1 <= nK - 2;  // second point of the pair is inside the loop range
#pragma unroll
    for (int di = 0; di < 2; di++) {
        const int64_t i = 2 * I + di;
        if (i > nI - 2 || i < ilo || i >= ihi) continue;
#pragma unroll
        for (int dj = 0; dj < 2; dj++) {
            const int64_t j = 2 * J + dj;
            if (j > nJ - 2) continue;
            const bool zero = i == 0 || j == 0;  // reference: eps = 0 where any index is 0
            double vx = (zero || K == 0) ? 0.0 : v[di][dj][0];
            double vy = zero ? 0.0 : v[di][dj][1];
            dpair *o = reinterpret_cast<dpair *>(out.p + out.at(i, j, 2 * K));
            if (CORRECT) {
                if (zero) continue;  // phi - 0 == phi
                dpair t = *o;
                if (K != 0) t.x = t.x - vx;
                if (k1) t.y = t.y - vy;
                *o = t;
            } else if (k1) {
                dpair t;
                t.x = vx;
                t.y = vy;
                *o = t;
            } else {
                out.p[out.at(i, j, 2 * K)] = vx;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_correct(double *__restrict__ phi,
                                                 const double *__restrict__ eps, int64_t count)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < count; p += stride)
        phi[p] = phi[p] - eps[p];
}

}  // namespace

// mg_set_exact_division(): every quotient by the division instruction (see div_d)
std::atomic<bool> g_exact_division{false};

void set_exact_division(bool on) { g_exact_division.store(on, std::memory_order_relaxed); }

Coef make_coef(double cx, double cy, double cz, double d)
{
    Coef c;
    c.cx = cx;
    c.cy = cy;
    c.cz = cz;
    c.d = d;
    c.rd = 1.0 / d;
    c.fast = 0;
    c.fb = c.fx = c.fy = c.fz = c.fu = 0.0;
    // the reciprocal form needs a positive d within 2^-100..2^100 (any real mesh coefficient)
    // whose significand is not all ones; div_d() bounds the numerator
    uint64_t bits;
    memcpy(&bits, &d, sizeof(bits));
    const uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
    const int expo = (int)((bits >> 52) & 0x7FF);
    c.fma_div = (!g_exact_division.load(std::memory_order_relaxed) && d > 0 && expo > 923 && expo < 1123 && frac != 0xFFFFFFFFFFFFFull) ? 1 : 0;
    return c;
}

int exp_knob(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

int64_t pitch_for(int64_t nk)
{
    static const int mode = exp_knob("MG_EXP_PITCH", 0);
    if (mode == 1 && nk >= 192) {  // an odd multiple of 256 bytes: consecutive rows start one channel apart
        int64_t p = (nk + 31) / 32;
        if (!(p & 1)) p++;
        return p * 32;
    }
    if (mode > 1 && nk >= 192) return ((nk + 1) & ~int64_t(1)) + mode;
    return (nk + 1) & ~int64_t(1);
}

Coef with_fast_smoother(Coef c, double omega)
{
    c.fast = 1;
    c.fb = omega * c.rd;
    c.fx = (omega * c.cx) * c.rd;
    c.fy = (omega * c.cy) * c.rd;
    c.fz = (omega * c.cz) * c.rd;
    c.fu = 1.0 - omega;
    return c;
}

namespace {

inline dim3 grid_for(int64_t nk, int64_t nj, int64_t ni)
{
    return dim3((unsigned)((nk + kBX - 1) / kBX), (unsigned)((nj + kBY - 1) / kBY), (unsigned)ni);
}

}  // namespace

int64_t residual_partials(const int64_t n[3], IRange r)
{
    dim3 g = grid_for(n[2], n[1], r.hi - r.lo);
    return (int64_t)g.x * g.y * g.z;
}

static int sumsq_blocks(int64_t count)
{
    int64_t b = (count + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int64_t sumsq_partials(int64_t count) { return sumsq_blocks(count); }

void repitch_in(hipStream_t s, View v, const double *dense, bool div)
{
    hipLaunchKernelGGL(k_repitch_in, grid_for(v.d[2], v.d[1], v.d[0]), dim3(kBX, kBY), 0, s, v, dense, div ? 1 : 0);
}

void repitch_out(hipStream_t s, double *dense, View v)
{
    hipLaunchKernelGGL(k_repitch_out, grid_for(v.d[2], v.d[1], v.d[0]), dim3(kBX, kBY), 0, s, dense, v);
}

void div_eps0(hipStream_t s, View v)
{
    hipLaunchKernelGGL(k_div_eps0, grid_for(v.d[2], v.d[1], v.d[0]), dim3(kBX, kBY), 0, s, v);
}

void smooth_colour(hipStream_t s, View u, View rhs, const int64_t n[3], Coef c, double omega,
                   int colour, IRange r)
{
    if (n[0] < 3 || n[1] < 3 || n[2] < 3) return;
    const int64_t lo = r.lo < 1 ? 1 : r.lo, hi = r.hi > n[0] - 1 ? n[0] - 1 : r.hi;
    if (hi <= lo) return;
    const int64_t half = (n[2] - 2 + 1) / 2;  // k positions of one colour in a row, at most
    if (c.fast)
        hipLaunchKernelGGL(k_smooth_colour<true>, grid_for(half, n[1] - 2, hi - lo), dim3(kBX, kBY), 0, s, u, rhs,
                           n[1], n[2], c, omega, colour, lo);
    else
        hipLaunchKernelGGL(k_smooth_colour<false>, grid_for(half, n[1] - 2, hi - lo), dim3(kBX, kBY), 0, s, u, rhs,
                           n[1], n[2], c, omega, colour, lo);
}

// Kernel choice for the smoother: the fused single-pass sweep pays once the level
// no longer fits the caches; tiny levels are launch-latency bound either way (measured: with
// the threshold at 64 rows the 513^3 cycle is 1 % slower).
static bool use_fused_sweep(const View &u) { return u.d[0] >= 4 && u.d[1] >= 128 && u.d[2] >= 128; }

// The marching kernels number their workgroups XCD-blocked (WgOrder); plain order cost the 513^3
// fine sweep 18 % more read traffic.
bool wg_order_blocked() { return true; }

// Plane ranges up to this length go to the tile kernel even where the marching kernel could run (the
// two planes next to a slab neighbour: a march over them is all lead-in planes and start-up).
// Measured on 32 planes of 257^2 / 513^2: the march 14 / 36 us, tiles 21 / 71 us; on 2 planes of
// 257^2: the march 8 us, tiles 4 us.  Planes of 129^2, the smallest the march takes: 40 planes 13.4
// (in pairs 11.2 per sweep) against 9.1 us, 24 planes 12.1 against 7.8, all 129 planes 16.1 per
// sweep in pairs against 19.7.
static int tile_max_planes(const View &u)
{
    static const int v = exp_knob("MG_EXP_TILE_PLANES", 2);
    static const int small = exp_knob("MG_EXP_TILE_PLANES_SMALL", 48);
    return u.d[1] * u.d[2] <= 136 * 136 && small > v ? small : v;
}
static bool use_tile_sweep() { static const int v = exp_knob("MG_EXP_NO_TILE", 0); return v == 0; }

bool sweep_is_out_of_place(const View &u) { return use_fused_sweep(u) || (use_tile_sweep() && u.d[0] >= 1); }

// Tiles take two sweeps per launch on planes up to this many points: there a launch is latency, and
// three launches for five sweeps beat five.
static int64_t tile_pair_max_area()
{
    static const int v = exp_knob("MG_EXP_TILE2_ROWS", 72);
    return (int64_t)v * v;
}

bool sweeps_in_pairs(const View &u, const View &alt, const View &rhs, const int64_t n[3], IRange r)
{
    const bool tile = alt.p && use_tile_sweep() && (!use_fused_sweep(u) || r.hi - r.lo <= tile_max_planes(u));
    const bool fused = !tile && alt.p && use_fused_sweep(u);
    // two sweeps per launch where the stored planes reach four beyond the range (always, for a
    // whole array; a slab with two halo planes exchanges between sweeps instead)
    const auto holds = [&](const View &v, int64_t lo, int64_t hi, int64_t top) {
        return v.i0 <= (lo < 0 ? 0 : lo) && v.i_end() >= (hi > top ? top : hi);
    };
    const bool by_tiles = tile && u.d[1] * u.d[2] <= tile_pair_max_area();
    return (fused || by_tiles) && holds(u, r.lo - 4, r.hi + 4, n[0]) && holds(rhs, r.lo - 3, r.hi + 3, n[0] - 1);
}

int smooth_rb(hipStream_t s, View *u, View *alt, View rhs, const int64_t n[3], Coef c,
              double omega, int sweeps, IRange r, IRange r2)
{
    if (n[0] < 3 || n[1] < 3 || n[2] < 3) return 0;
    const bool pair = r2.hi > r2.lo;
    if (pair && !(sweeps == 1 && alt && alt->p && sweep_is_out_of_place(*u) && r2.hi - r2.lo == r.hi - r.lo &&
                  r2.lo >= r.hi))
        return -1;  // a second range needs an out-of-place kernel and one sweep
    const bool whole = !pair && u->i0 == 0 && r.lo == 0 && r.hi == u->d[0] && rhs.i0 == 0;
    // small enough for one workgroup: all sweeps in a single launch, u in LDS, <= 4 points per
    // colour per thread in registers
    const size_t lds_bytes = sizeof(double) * smooth_lds_doubles(u->d[0], u->d[1], u->d[2]);
    const int64_t per_colour = (n[0] - 2) * (n[1] - 2) * ((n[2] - 1) / 2);
    static const bool lds_ok = [] {
        bool ok = true;
        for (const void *f : {(const void *)k_smooth_lds<2, 256, false>, (const void *)k_smooth_lds<2, 1024, false>,
                              (const void *)k_smooth_lds<4, 1024, false>, (const void *)k_smooth_lds<2, 256, true>,
                              (const void *)k_smooth_lds<2, 1024, true>, (const void *)k_smooth_lds<4, 1024, true>})
            ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsSmoothMaxBytes) == hipSuccess;
        return ok;
    }();
    if (whole && lds_ok && lds_bytes <= (size_t)kLdsSmoothMaxBytes && per_colour <= 4 * 1024 && sweeps > 0) {
        // Thread count by measurement (50 sweeps; 256 / 512 / 1024 threads): 365 points per colour
        // 31 / 37 / 53 us, 1372 points 98 / 74 / 68 us, 2916 points - / 148 / 125 us.  The phases of a
        // tiny level are latency (fewer waves meet at the barriers), those of a larger one throughput.
#define LDS_LAUNCH(P, NT)                                                                                             \
    do {                                                                                                              \
        if (c.fast)                                                                                                   \
            hipLaunchKernelGGL((k_smooth_lds<P, NT, true>), dim3(1), dim3(NT), lds_bytes, s, *u, rhs, (int)n[0],      \
                               (int)n[1], (int)n[2], c, omega, sweeps);                                               \
        else                                                                                                          \
            hipLaunchKernelGGL((k_smooth_lds<P, NT, false>), dim3(1), dim3(NT), lds_bytes, s, *u, rhs, (int)n[0],     \
                               (int)n[1], (int)n[2], c, omega, sweeps);                                               \
    } while (0)
        if (per_colour <= 2 * 256) LDS_LAUNCH(2, 256);
        else if (per_colour <= 2 * 1024) LDS_LAUNCH(2, 1024);
        else LDS_LAUNCH(4, 1024);
#undef LDS_LAUNCH
        return 1;
    }
    const bool tile = alt && alt->p && use_tile_sweep() && (!use_fused_sweep(*u) || r.hi - r.lo <= tile_max_planes(*u));
    const bool fused = !tile && alt && alt->p && use_fused_sweep(*u);
    const bool pairs = !pair && alt && sweeps_in_pairs(*u, *alt, rhs, n, r);
    int launches = 0;
    for (int sw = 0; sw < sweeps; sw++) {
        launches++;
        if (pairs && sw + 1 < sweeps) {
            if (tile) sweep_rb_tile2(s, *u, *alt, rhs, n, c, omega, r);
            else sweep_rb_fused2(s, *u, *alt, rhs, n, c, omega, r);
            View t = *u;
            *u = *alt;
            *alt = t;
            sw++;
        } else if (fused || tile) {
            if (tile) sweep_rb_tile(s, *u, *alt, rhs, n, c, omega, r, r2);
            else sweep_rb_fused(s, *u, *alt, rhs, n, c, omega, r, r2);
            View t = *u;
            *u = *alt;
            *alt = t;
        } else {
            // Red also on the plane next to the range on either side (as far as its own
            // neighbours are stored): in a slab partition those are halo planes whose red
            // values the black update below needs, and computing them here instead of
            // receiving them lets one halo exchange per sweep suffice.  For a whole array
            // the wider range clips back to 1..n-2.
            IRange red{r.lo - 1, r.hi + 1};
            if (red.lo < u->i0 + 1) red.lo = u->i0 + 1;
            if (red.hi > u->i_end() - 1) red.hi = u->i_end() - 1;
            smooth_colour(s, *u, rhs, n, c, omega, 0, red);
            smooth_colour(s, *u, rhs, n, c, omega, 1, r);
            launches++;
        }
    }
    return launches;
}

bool onload_folds(const View &u, bool correct)
{
    // On-load prolongation pays on big levels only (measured at 129^3: 35 us fused against
    // 23 + 5 us separate; at 257^3: 125 against 107 + 27); the fine-grid correction always pays.
    const bool big = correct || (u.d[1] >= 256 && u.d[2] >= 256);
    return big && use_fused_sweep(u);
}

int prolong_smooth_rb(hipStream_t s, View ec, bool correct, View *u, View *alt, View rhs,
                      const int64_t n[3], Coef c, double omega, int sweeps, IRange r, IRange r2, bool outside_zero)
{
    const bool pair = r2.hi > r2.lo;
    if (pair && !(sweeps == 1 && alt && alt->p && sweep_is_out_of_place(*u) && r2.hi - r2.lo == r.hi - r.lo &&
                  r2.lo >= r.hi))
        return -1;
    if (sweeps > 0 && alt && alt->p && onload_folds(*u, correct)) {
        sweep_rb_fused_onload(s, *u, *alt, rhs, ec, correct, n, c, omega, r, r2, outside_zero);
        View t = *u;
        *u = *alt;
        *alt = t;
        return 1 + smooth_rb(s, u, alt, rhs, n, c, omega, sweeps - 1, r);
    }
    // separate launches: the operation must reach every stored plane the first sweep reads
    // (planes r.lo-2 .. r.hi+1; in a slab those are halo planes, whose coarse neighbours the
    // caller keeps current)
    int launches = 0;
    for (int part = 0; part < (pair ? 2 : 1); part++) {
        const IRange q = part ? r2 : r;
        IRange pr{q.lo - 2, q.hi + 2};
        if (part && pr.lo < r.hi + 2) pr.lo = r.hi + 2;  // planes the first part already covered
        if (pr.lo < u->i0) pr.lo = u->i0;
        if (pr.hi > u->i_end()) pr.hi = u->i_end();
        if (pr.hi > n[0]) pr.hi = n[0];
        if (correct) prolong_correct(s, ec, *u, n, pr);
        else prolong(s, ec, *u, n, pr);
        launches++;
    }
    return launches + smooth_rb(s, u, alt, rhs, n, c, omega, sweeps, r, r2);
}

void residual(hipStream_t s, View phi, View b, View R, const int64_t n[3], Coef c, double *partials,
              IRange r)
{
    hipLaunchKernelGGL(k_residual, grid_for(n[2], n[1], r.hi - r.lo), dim3(kBX, kBY), 0, s, phi, b, R,
                       n[0], c, partials, r.lo);
}

void sum_partials(hipStream_t s, const double *partials, int64_t count, double *out)
{
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, s, partials, count, out);
}

void sumsq(hipStream_t s, const double *x, int64_t count, double *partials)
{
    hipLaunchKernelGGL(k_sumsq, dim3(sumsq_blocks(count)), dim3(256), 0, s, x, count, partials);
}

void restrict27(hipStream_t s, View Rf, View Rc, const int64_t nf[3], IRange rc)
{
    (void)nf;
    if (rc.hi <= rc.lo) return;
    hipLaunchKernelGGL(k_restrict27, grid_for(Rc.d[2], Rc.d[1], rc.hi - rc.lo), dim3(kBX, kBY), 0, s, Rf,
                       Rc, rc.lo);
}

void restrict7(hipStream_t s, View Rf, View Rc, const int64_t nf[3], IRange rc)
{
    (void)nf;
    if (rc.hi <= rc.lo) return;
    hipLaunchKernelGGL(k_restrict7, grid_for(Rc.d[2], Rc.d[1], rc.hi - rc.lo), dim3(kBX, kBY), 0, s, Rf, Rc,
                       rc.lo);
}

static void launch_prolong(hipStream_t s, bool correct, View ec, View out, const int64_t n[3], IRange r)
{
    if (r.hi <= r.lo) return;
    const int64_t cells = ((r.hi - 1) >> 1) - (r.lo >> 1) + 1;
    const dim3 g = grid_for(n[2] / 2, n[1] / 2, cells);
    if (correct)
        hipLaunchKernelGGL(k_prolong<true>, g, dim3(kBX, kBY), 0, s, ec, out, n[0], n[1], n[2], r.lo, r.hi);
    else
        hipLaunchKernelGGL(k_prolong<false>, g, dim3(kBX, kBY), 0, s, ec, out, n[0], n[1], n[2], r.lo, r.hi);
}

void prolong(hipStream_t s, View ec, View ef, const int64_t n[3], IRange r)
{
    launch_prolong(s, false, ec, ef, n, r);
}

void prolong_correct(hipStream_t s, View ec, View phi, const int64_t n[3], IRange r)
{
    launch_prolong(s, true, ec, phi, n, r);
}

void correct(hipStream_t s, double *phi, const double *eps, int64_t count)
{
    int64_t b = (count + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(k_correct, dim3((unsigned)b), dim3(256), 0, s, phi, eps, count);
}

}  // namespace mgk
