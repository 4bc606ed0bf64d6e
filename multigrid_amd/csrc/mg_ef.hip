// E = -grad(phi) on the device-resident potential (reference cpp/PotentialSolver.cpp:418-458):
// central differences inside, 3-point one-sided differences on the faces, per axis.
//
// One pass over phi: a 256-thread workgroup owns a 4 x 64 (j x k) window and marches along i
// with the i-1 / i / i+1 values of its column in registers; the j and k neighbours of the
// current plane are plain global loads that hit L1/L2 (the same lines are the centre values of
// neighbouring lanes and rows).  The three components of a node are adjacent in the output
// ([i][j][k][3], the layout of the reference's Field_<double3>), so a lane's own results are
// 24 bytes apart; they go through an LDS row image and leave as contiguous 8-byte stores.
#include "mg_kernels.h"

namespace mgk {

namespace {

constexpr int EF_ROWS = 4;
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

__global__ __launch_bounds__(256) void k_compute_ef(View phi, double *__restrict__ ef, int64_t ef_i0,
                                                    int64_t ni, Coef hx, Coef hy, Coef hz, IRange r, EfGeom g)
{
    __shared__ double tile[2][EF_ROWS][EF_COLS * 3];
    const int64_t nj = phi.d[1], nk = phi.d[2];
    const int chunk = blockIdx.x % g.chunks;
    const int win = blockIdx.x / g.chunks;
    const int kt = win % g.ktiles, jt = win / g.ktiles;
    const int row = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)jt * EF_ROWS + row, k = (int64_t)kt * EF_COLS + lane;
    const bool valid = j < nj && k < nk;
    const int64_t jc = clampi(j, 0, nj - 1), kc = clampi(k, 0, nk - 1);
    const int64_t lo = r.lo + chunk * g.ci, hi = (lo + g.ci < r.hi) ? lo + g.ci : r.hi;
    if (lo >= hi) return;  // uniform per workgroup

    const int64_t jm1 = clampi(jc - 1, 0, nj - 1), jp1 = clampi(jc + 1, 0, nj - 1);
    const int64_t km1 = clampi(kc - 1, 0, nk - 1), kp1 = clampi(kc + 1, 0, nk - 1);
    // second neighbour, needed on the faces only: j+2 at j=0, j-2 at j=nj-1 (same for k)
    const int64_t j2 = (jc == 0) ? 2 : nj - 3, k2 = (kc == 0) ? 2 : nk - 3;
    const bool jface = jc == 0 || jc == nj - 1, kface = kc == 0 || kc == nk - 1;

    const double *p = phi.p;
    // prefetches past the range stay inside the planes the (possibly slab) view stores
    const int64_t plo = phi.i0, phi_last = phi.i_end() - 1;
    double um1 = p[phi.at(clampi(lo - 1, plo, phi_last), jc, kc)];
    double u0 = p[phi.at(lo, jc, kc)];
    double up1 = p[phi.at(clampi(lo + 1, plo, phi_last), jc, kc)];
    const int64_t row_doubles = nk * 3;
    const int64_t seg0 = (int64_t)kt * EF_COLS * 3;                       // first double of this window in a row
    const int64_t seg_len = ((nk - (int64_t)kt * EF_COLS < EF_COLS) ? nk - (int64_t)kt * EF_COLS : EF_COLS) * 3;

    int buf = 0;
    for (int64_t i = lo; i < hi; i++, buf ^= 1) {
        const double unext = p[phi.at(clampi(i + 2, plo, phi_last), jc, kc)];  // becomes up1 next plane
        const double vjm = p[phi.at(i, jm1, kc)], vjp = p[phi.at(i, jp1, kc)];
        const double vkm = p[phi.at(i, jc, km1)], vkp = p[phi.at(i, jc, kp1)];
        double vi2 = 0, vj2 = 0, vk2 = 0;
        if (i == 0) vi2 = unext;                                           // plane 2
        else if (i == ni - 1) vi2 = p[phi.at(ni - 3, jc, kc)];
        if (jface) vj2 = p[phi.at(i, j2, kc)];
        if (kface) vk2 = p[phi.at(i, jc, k2)];

        const double ex = ef_axis(vi2, um1, u0, up1, vi2, i, ni - 1, hx);
        const double ey = ef_axis(vj2, vjm, u0, vjp, vj2, jc, nj - 1, hy);
        const double ez = ef_axis(vk2, vkm, u0, vkp, vk2, kc, nk - 1, hz);

        double *t = &tile[buf][row][0];
        t[3 * lane + 0] = ex;
        t[3 * lane + 1] = ey;
        t[3 * lane + 2] = ez;
        __syncthreads();
        if (j < nj) {
            double *out = ef + ((i - ef_i0) * nj + j) * row_doubles + seg0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int e = lane + EF_COLS * c;
                if (e < seg_len) out[e] = t[e];
            }
        }
        um1 = u0;
        u0 = up1;
        up1 = unext;
    }
    (void)valid;
}

}  // namespace

void compute_ef(hipStream_t s, View phi, double *ef, int64_t ef_i0, const int64_t n[3], const double dh[3],
                IRange r)
{
    if (r.hi <= r.lo) return;
    EfGeom g;
    g.ktiles = (int)((n[2] + EF_COLS - 1) / EF_COLS);
    g.jtiles = (int)((n[1] + EF_ROWS - 1) / EF_ROWS);
    const int64_t planes = r.hi - r.lo;
    g.chunks = pick_chunks((int64_t)g.ktiles * g.jtiles, planes, 3, 1, 2048);
    g.ci = (planes + g.chunks - 1) / g.chunks;
    const Coef hx = make_coef(0, 0, 0, 2 * dh[0]), hy = make_coef(0, 0, 0, 2 * dh[1]),
               hz = make_coef(0, 0, 0, 2 * dh[2]);
    const unsigned grid = (unsigned)((int64_t)g.ktiles * g.jtiles * g.chunks);
    hipLaunchKernelGGL(k_compute_ef, dim3(grid), dim3(256), 0, s, phi, ef, ef_i0, n[0], hx, hy, hz, r, g);
}

}  // namespace mgk
