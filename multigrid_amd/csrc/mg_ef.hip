// E = -grad(phi) on the device-resident potential (reference cpp/PotentialSolver.cpp:418-458):
// central differences inside, 3-point one-sided differences on the faces, per axis.
//
// One pass over phi: a 256-thread workgroup owns a 16 x 64 (j x k) window and marches along i;
// a lane owns four vertically adjacent nodes of one column with their i-1 / i / i+1 values in
// registers, so most j neighbours are its own registers; the rows above and below its block and
// the k neighbours are plain global loads that hit L1/L2 (the same lines are the centre values
// of neighbouring lanes and waves).  Workgroups are numbered XCD-blocked (WgOrder) so that the
// halo rows shared with neighbouring windows are fetched once per L2.  The three components of
// a node are adjacent in the output ([i][j][k][3], the layout of the reference's
// Field_<double3>), so a lane's own results are 24 bytes apart; each wave passes a row through
// its own LDS row image and stores it as contiguous 8-byte words.
#include "mg_kernels.h"

namespace mgk {

namespace {

constexpr int EF_COLS = 64;

struct EfGeom {
    int ktiles, jtiles, chunks;
    int64_t ci;  // planes per chunk
};

__device__ __forceinline__ int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One axis: m2..p2 are the values at offsets -2..+2 along the axis (only the ones the case needs
// are meaningful), pos / last the node's index and the last index on the axis.
__device__ __forceinline__ double ef_axis(double m2, double m1, double c0, double p1, double p2,
                                          int64_t pos, int64_t last, const Coef &two_h)
{
    double num;
    if (pos == 0) num = -(-3 * c0 + 4 * p1 - p2);
    else if (pos == last) num = -(m2 - 4 * m1 + 3 * c0);
    else num = -(p1 - m1);
    return div_d(num, two_h);
}

// EF_RPT: adjacent rows per lane (window = 4 waves x EF_RPT rows)
template <int EF_RPT>
__global__ __launch_bounds__(256) void k_compute_ef(View phi, double *__restrict__ ef, int64_t ef_i0,
                                                    int64_t ni, Coef hx, Coef hy, Coef hz, IRange r, EfGeom g,
                                                    WgOrder order)
{
    constexpr int EF_ROWS = 4 * EF_RPT;
    __shared__ double tile[4][EF_COLS * 3];  // one output row image per wave
    const int item = order.item(blockIdx.x);
    if (item < 0) return;  // grid padding; uniform per workgroup
    const int64_t nj = phi.d[1], nk = phi.d[2];
    const int nwin = g.ktiles * g.jtiles;
    const int chunk = item / nwin, win = item % nwin;
    const int kt = win % g.ktiles, jt = win / g.ktiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t jb = (int64_t)jt * EF_ROWS + wave * EF_RPT;  // first row of this lane's block
    const int64_t k = (int64_t)kt * EF_COLS + lane;
    const int64_t kc = clampi(k, 0, nk - 1);
    const int64_t lo = r.lo + chunk * g.ci, hi = (lo + g.ci < r.hi) ? lo + g.ci : r.hi;
    if (lo >= hi) return;  // uniform per workgroup

    const int64_t km1 = clampi(kc - 1, 0, nk - 1), kp1 = clampi(kc + 1, 0, nk - 1);
    const int64_t k2 = (kc == 0) ? 2 : nk - 3;  // second neighbour, used on the k faces only
    const bool kface = kc == 0 || kc == nk - 1;
    int64_t jr[EF_RPT];  // clamped row indices (rows past the array repeat the last one; not stored)
#pragma unroll
    for (int q = 0; q < EF_RPT; q++) jr[q] = clampi(jb + q, 0, nj - 1);
    const int64_t jabove = clampi(jb - 1, 0, nj - 1), jbelow = clampi(jb + EF_RPT, 0, nj - 1);

    const double *p = phi.p;
    // prefetches past the range stay inside the planes the (possibly slab) view stores
    const int64_t plo = phi.i0, phi_last = phi.i_end() - 1;
    // Everything a plane needs is requested one plane ahead, so no load is waited for in the
    // iteration that issues it: centre values two planes ahead (they are also the i+1 neighbour),
    // the rows above and below the block, and per row the one k neighbour the wave does not hold
    // itself (lane 0: k-1, lane 63: k+1; the other lanes get theirs by shuffle from the centres).
    const int64_t kedge = lane == 0 ? km1 : (lane == 63 ? kp1 : kc);
    double um1[EF_RPT], u0[EF_RPT], up1[EF_RPT], edge[EF_RPT];
#pragma unroll
    for (int q = 0; q < EF_RPT; q++) {
        um1[q] = p[phi.at(clampi(lo - 1, plo, phi_last), jr[q], kc)];
        u0[q] = p[phi.at(lo, jr[q], kc)];
        up1[q] = p[phi.at(clampi(lo + 1, plo, phi_last), jr[q], kc)];
        edge[q] = p[phi.at(lo, jr[q], kedge)];
    }
    double above = p[phi.at(lo, jabove, kc)], below = p[phi.at(lo, jbelow, kc)];
    const int64_t row_doubles = nk * 3;
    const int64_t seg0 = (int64_t)kt * EF_COLS * 3;  // first double of this window in an output row
    const int64_t seg_len = ((nk - (int64_t)kt * EF_COLS < EF_COLS) ? nk - (int64_t)kt * EF_COLS : EF_COLS) * 3;
    double *t = &tile[wave][0];

    for (int64_t i = lo; i < hi; i++) {
        double unext[EF_RPT], edge_next[EF_RPT];
        const int64_t in1 = clampi(i + 1, plo, phi_last);
#pragma unroll
        for (int q = 0; q < EF_RPT; q++) {
            unext[q] = p[phi.at(clampi(i + 2, plo, phi_last), jr[q], kc)];  // becomes up1 next plane
            edge_next[q] = p[phi.at(in1, jr[q], kedge)];
        }
        const double above_next = p[phi.at(in1, jabove, kc)], below_next = p[phi.at(in1, jbelow, kc)];
#pragma unroll
        for (int q = 0; q < EF_RPT; q++) {
            const int64_t j = jb + q;
            const double vjm = q == 0 ? above : u0[q > 0 ? q - 1 : 0];
            const double vjp = q == EF_RPT - 1 ? below : u0[q < EF_RPT - 1 ? q + 1 : q];
            const double left = __shfl_up(u0[q], 1, 64), right = __shfl_down(u0[q], 1, 64);
            // at the array's own k faces the clamped neighbour is not used (one-sided formula)
            const double vkm = lane == 0 ? edge[q] : left;
            const double vkp = lane == 63 ? edge[q] : right;
            double vi2 = 0, vj2 = 0, vk2 = 0;
            if (i == 0) vi2 = unext[q];  // plane 2
            else if (i == ni - 1) vi2 = p[phi.at(ni - 3, jr[q], kc)];
            if (j == 0 || j == nj - 1) vj2 = p[phi.at(i, j == 0 ? 2 : nj - 3, kc)];
            if (kface) vk2 = p[phi.at(i, jr[q], k2)];
            const double ex = ef_axis(vi2, um1[q], u0[q], up1[q], vi2, i, ni - 1, hx);
            const double ey = ef_axis(vj2, vjm, u0[q], vjp, vj2, jr[q], nj - 1, hy);
            const double ez = ef_axis(vk2, vkm, u0[q], vkp, vk2, kc, nk - 1, hz);
            // LDS operations of a wave execute in order; the fences only keep the compiler from
            // moving the row image's reads above its writes (and the next row's writes above them)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            t[3 * lane + 0] = ex;
            t[3 * lane + 1] = ey;
            t[3 * lane + 2] = ez;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (j < nj) {  // wave-uniform
                double *out = ef + ((i - ef_i0) * nj + j) * row_doubles + seg0;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int e = lane + EF_COLS * c;
                    if (e < seg_len) out[e] = t[e];
                }
            }
        }
        above = above_next;
        below = below_next;
#pragma unroll
        for (int q = 0; q < EF_RPT; q++) edge[q] = edge_next[q];
#pragma unroll
        for (int q = 0; q < EF_RPT; q++) {
            um1[q] = u0[q];
            u0[q] = up1[q];
            up1[q] = unext[q];
        }
    }
}

}  // namespace

void compute_ef(hipStream_t s, View phi, double *ef, int64_t ef_i0, const int64_t n[3], const double dh[3],
                IRange r)
{
    if (r.hi <= r.lo) return;
    // measured at 513^3: 1, 2 and 4 rows per lane run within 5 % of each other (1.49 / 1.42 / 1.42 ms);
    // storing the components directly (24-byte stride across a wave) instead of through the LDS row
    // image costs 18 %
    constexpr int RPT = 4;
    const int rows = 4 * RPT;
    EfGeom g;
    g.ktiles = (int)((n[2] + EF_COLS - 1) / EF_COLS);
    g.jtiles = (int)((n[1] + rows - 1) / rows);
    const int64_t planes = r.hi - r.lo;
    const bool blocked = wg_order_blocked();
    g.chunks = pick_chunks((int64_t)g.ktiles * g.jtiles, planes, 3, 1, 2048, !blocked);
    g.ci = (planes + g.chunks - 1) / g.chunks;
    const WgOrder order{g.ktiles * g.jtiles * g.chunks, blocked ? 1 : 0};
    const Coef hx = make_coef(0, 0, 0, 2 * dh[0]), hy = make_coef(0, 0, 0, 2 * dh[1]),
               hz = make_coef(0, 0, 0, 2 * dh[2]);
    hipLaunchKernelGGL((k_compute_ef<RPT>), dim3(order.grid()), dim3(256), 0, s, phi, ef, ef_i0, n[0], hx, hy, hz,
                       r, g, order);
}

}  // namespace mgk
