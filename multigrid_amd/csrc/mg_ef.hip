// E = -grad(phi) on the device-resident potential (reference cpp/PotentialSolver.cpp:418-458):
// central differences inside, 3-point one-sided differences on the faces, per axis.
//
// One pass over phi: a 256-thread workgroup owns a (4*JR) x 128 (j x k) window and marches along
// i.  A lane owns a PAIR of adjacent columns (16-byte loads) in JR vertically adjacent rows, with
// their i-1 / i / i+1 values in registers: the i neighbours and most j neighbours are its own
// registers, the k neighbours across a pair boundary come from the adjacent lane by a DPP wave
// shift, and only the rows above and below a wave's block and the pair beside the wave's span are
// extra loads (cache hits: they are the centre values of neighbouring waves).  A 128-column row
// segment is eight whole cache lines plus at most one shared one, a 4*JR-row window re-reads two
// halo rows: that is what keeps the read traffic near the 8 bytes per node it has to be.
// Workgroups are numbered XCD-blocked (WgOrder) so that the lines shared with neighbouring windows
// are fetched once per L2.  The three components of a node are adjacent in the output
// ([i][j][k][3], the layout of the reference's Field_<double3>): each wave passes a row through
// its own LDS row image and stores it as contiguous 16-byte words laid over the output's 128-byte
// lines (a row of the output starts on an 8-byte boundary: the doubles in front of the first line
// boundary go out as one short store).
//
// Grids of 2^m + 1 nodes leave one column over after the last full window; a window for it would
// be a workgroup doing 1/128 of a window's work (20 % more workgroups at 513 columns).  The
// marching windows therefore cover the columns up to the last multiple of 128 and extra
// workgroups of the same launch, one per plane, the few columns left, node by node.
#include "mg_kernels.h"

namespace mgk {

namespace {

constexpr int EF_COLS = 128;

struct alignas(16) d2 { double x, y; };
typedef double v2d __attribute__((ext_vector_type(2)));

struct EfGeom {
    int ktiles, jtiles, chunks;
    int64_t ci;  // planes per chunk
};

__device__ __forceinline__ int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Value of lane - 1 (lane 0 keeps its own) / of lane + 1 (lane 63 keeps its own).
__device__ __forceinline__ double lane_left(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_right(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// One axis: m2..p2 are the values at offsets -2..+2 along the axis (only the ones the case needs
// are meaningful), pos / last the node's index and the last index on the axis.
__device__ __forceinline__ double ef_num(double m2, double m1, double c0, double p1, double p2,
                                         int64_t pos, int64_t last)
{
    if (pos == 0) return -(-3 * c0 + 4 * p1 - p2);
    if (pos == last) return -(m2 - 4 * m1 + 3 * c0);
    return -(p1 - m1);
}
__device__ __forceinline__ double ef_axis(double m2, double m1, double c0, double p1, double p2,
                                          int64_t pos, int64_t last, const Coef &two_h)
{
    return div_d(ef_num(m2, m1, c0, p1, p2, pos, last), two_h);
}

// The same operator node by node for columns [kbeg, nk) of plane i: the few columns the marching
// windows leave over.  Every operand is a plain load; the workgroup's threads share the plane's
// nodes.
__device__ __forceinline__ void ef_columns_of_plane(const View &phi, double *__restrict__ ef, int64_t ef_i0, int64_t ni,
                                                    int64_t kbeg, const Coef &hx, const Coef &hy, const Coef &hz,
                                                    int64_t i)
{
    const int64_t nj = phi.d[1], nk = phi.d[2];
    const int64_t ncol = nk - kbeg;
    const double *p = phi.p;
    auto at = [&](int64_t a, int64_t b, int64_t c) { return p[phi.at(a, b, c)]; };
    for (int64_t t = threadIdx.x; t < nj * ncol; t += blockDim.x) {
        const int64_t k = kbeg + t % ncol, j = t / ncol;
        const double c0 = at(i, j, k);
        // offsets -2..+2 along each axis, clamped where the one-sided case does not read them
        auto axis = [&](int a, int64_t pos, int64_t last, const Coef &h) {
            auto v = [&](int64_t s) {
                const int64_t q = clampi(pos + s, 0, last);
                return a == 0 ? at(q, j, k) : (a == 1 ? at(i, q, k) : at(i, j, q));
            };
            return ef_axis(v(-2), v(-1), c0, v(1), v(2), pos, last, h);
        };
        double *out = ef + (((i - ef_i0) * nj + j) * nk + k) * 3;
        out[0] = axis(0, i, ni - 1, hx);
        out[1] = axis(1, j, nj - 1, hy);
        out[2] = axis(2, k, nk - 1, hz);
    }
}

// JR: adjacent rows per lane (window = 4 waves x JR rows).  The first windows x chunks workgroups
// march and produce columns [0, kmain); each workgroup after them takes columns [kmain, nk) of one
// plane, in the same launch, so that this latency-bound remainder runs beside the windows instead
// of behind them.
template <int JR>
__global__ __launch_bounds__(256) void k_compute_ef(View phi, double *__restrict__ ef, int64_t ef_i0,
                                                    int64_t ni, int64_t kmain, Coef hx, Coef hy, Coef hz, IRange r,
                                                    EfGeom g, WgOrder order)
{
    constexpr int ROWS = 4 * JR;
    __shared__ __align__(16) double tile[4][JR][EF_COLS * 3];  // per wave: one output row image per owned row
    if (blockIdx.x >= order.grid()) {  // uniform per workgroup
        // dealt round-robin to the XCDs like any run of workgroup indices, and last: they fill the
        // slots the final, partly filled round of marching workgroups leaves free
        ef_columns_of_plane(phi, ef, ef_i0, ni, kmain, hx, hy, hz, r.lo + (blockIdx.x - order.grid()));
        return;
    }
    const int item = order.item(blockIdx.x);
    if (item < 0) return;  // grid padding; uniform per workgroup
    const int64_t nj = phi.d[1], nk = phi.d[2];
    const int nwin = g.ktiles * g.jtiles;
    const int chunk = item / nwin, win = item % nwin;
    const int kt = win % g.ktiles, jt = win / g.ktiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t jb = (int64_t)jt * ROWS + wave * JR;  // first row of this lane's block
    const int64_t lo = r.lo + chunk * g.ci, hi = (lo + g.ci < r.hi) ? lo + g.ci : r.hi;
    if (lo >= hi) return;  // uniform per workgroup

    // The lane's pair of columns (kx, kx+1).  Rows are stored with an even pitch, so the pair that
    // starts at the last even column is always loadable; lanes past it repeat it (never stored).
    const int64_t kpair_max = (nk - 1) & ~int64_t(1);
    const int64_t k0 = (int64_t)kt * EF_COLS + 2 * lane;
    const int64_t kx = k0 < kpair_max ? k0 : kpair_max;
    // the pair beside the wave's span: lane 0 needs columns kx-2, kx-1, lane 63 columns kx+2, kx+3
    const int64_t ke = lane == 0 ? (kx >= 2 ? kx - 2 : 0) : (lane == 63 ? (kx + 2 < kpair_max ? kx + 2 : kpair_max) : kx);
    int64_t jr[JR];  // clamped row indices (rows past the array repeat the last one; not stored)
#pragma unroll
    for (int q = 0; q < JR; q++) jr[q] = clampi(jb + q, 0, nj - 1);
    const int64_t jabove = clampi(jb - 1, 0, nj - 1), jbelow = clampi(jb + JR, 0, nj - 1);

    const double *p = phi.p;
    auto pair_at = [&](int64_t i, int64_t j, int64_t k) -> d2 { return *reinterpret_cast<const d2 *>(p + phi.at(i, j, k)); };
    // prefetches past the range stay inside the planes the (possibly slab) view stores
    const int64_t plo = phi.i0, phi_last = phi.i_end() - 1;
    // Everything a plane needs is requested one plane ahead, so no load is waited for in the
    // iteration that issues it: centre pairs two planes ahead (they are also the i+1 neighbour),
    // the rows above and below the block, and per row the pair beside the wave's span.
    d2 um1[JR], u0[JR], up1[JR], edge[JR];
#pragma unroll
    for (int q = 0; q < JR; q++) {
        um1[q] = pair_at(clampi(lo - 1, plo, phi_last), jr[q], kx);
        u0[q] = pair_at(lo, jr[q], kx);
        up1[q] = pair_at(clampi(lo + 1, plo, phi_last), jr[q], kx);
        edge[q] = pair_at(lo, jr[q], ke);
    }
    d2 above = pair_at(lo, jabove, kx), below = pair_at(lo, jbelow, kx);
    const int64_t row_doubles = nk * 3;
    const int64_t seg0 = (int64_t)kt * EF_COLS * 3;  // first double of this window in an output row
    const int64_t cols_here = kmain - (int64_t)kt * EF_COLS < EF_COLS ? kmain - (int64_t)kt * EF_COLS : EF_COLS;
    const int seg_len = (int)(cols_here * 3);
    double *t = &tile[wave][0][0];
    const bool x_first = kx == 0, x_last = kx == nk - 1, y_last = kx + 1 == nk - 1;
    const bool k_face = x_first || x_last || y_last;

    for (int64_t i = lo; i < hi; i++) {
        d2 unext[JR], edge_next[JR];
        const int64_t in1 = clampi(i + 1, plo, phi_last);
#pragma unroll
        for (int q = 0; q < JR; q++) {
            unext[q] = pair_at(clampi(i + 2, plo, phi_last), jr[q], kx);  // becomes up1 next plane
            edge_next[q] = pair_at(in1, jr[q], ke);
        }
        const d2 above_next = pair_at(in1, jabove, kx), below_next = pair_at(in1, jbelow, kx);
#pragma unroll
        for (int q = 0; q < JR; q++) {
            const int64_t j = jb + q;
            const d2 u = u0[q];
            const d2 vjm = q == 0 ? above : u0[q > 0 ? q - 1 : 0];
            const d2 vjp = q == JR - 1 ? below : u0[q < JR - 1 ? q + 1 : q];
            // columns kx-2, kx-1 (the left lane's pair) and kx+2 (the right lane's first element)
            double l2 = lane_left(u.x), l1 = lane_left(u.y), r1 = lane_right(u.x);
            if (lane == 0) { l2 = edge[q].x; l1 = edge[q].y; }
            if (lane == 63) r1 = edge[q].x;
            // Numerators of the two nodes per axis: the central difference everywhere, replaced by
            // the one-sided form on a face.  A face in i or j is a whole plane or a whole row of
            // the wave (scalar branches), a face in k is one lane of the waves at either end of a
            // row: no node inside the grid evaluates a one-sided formula only to discard it.
            double nx[2], ny[2], nz[2], qx[2], qy[2], qz[2];
            nx[0] = -(up1[q].x - um1[q].x);
            nx[1] = -(up1[q].y - um1[q].y);
            ny[0] = -(vjp.x - vjm.x);
            ny[1] = -(vjp.y - vjm.y);
            nz[0] = -(u.y - l1);
            nz[1] = -(r1 - u.x);
            if (i == 0 || i == ni - 1) {
                const d2 vi2 = i == 0 ? unext[q] /* plane 2 */ : pair_at(ni - 3, jr[q], kx);
                nx[0] = ef_num(vi2.x, um1[q].x, u.x, up1[q].x, vi2.x, i, ni - 1);
                nx[1] = ef_num(vi2.y, um1[q].y, u.y, up1[q].y, vi2.y, i, ni - 1);
            }
            if (j == 0 || j == nj - 1) {
                const d2 vj2 = pair_at(i, j == 0 ? 2 : nj - 3, kx);
                ny[0] = ef_num(vj2.x, vjm.x, u.x, vjp.x, vj2.x, j, nj - 1);
                ny[1] = ef_num(vj2.y, vjm.y, u.y, vjp.y, vj2.y, j, nj - 1);
            }
            if (k_face) {
                // the first column's second neighbour is column 2 (r1), the last column's are the
                // two columns before it
                nz[0] = ef_num(l2, l1, u.x, u.y, r1, x_first ? 0 : (x_last ? nk - 1 : 1), nk - 1);
                nz[1] = ef_num(l1, u.x, u.y, r1, 0.0, y_last ? nk - 1 : 1, nk - 1);
            }
            div_d_n<2>(nx, hx, qx);
            div_d_n<2>(ny, hy, qy);
            div_d_n<2>(nz, hz, qz);
            const d2 ex{qx[0], qx[1]}, ey{qy[0], qy[1]}, ez{qz[0], qz[1]};
            d2 *t2 = reinterpret_cast<d2 *>(t + q * (EF_COLS * 3) + 6 * lane);
            t2[0] = d2{ex.x, ey.x};
            t2[1] = d2{ez.x, ex.y};
            t2[2] = d2{ey.y, ez.y};
        }
        // LDS operations of a wave execute in order; the fence only keeps the compiler from moving
        // the row images' reads above their writes (and the next plane's writes above the reads)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int q = 0; q < JR; q++) {
            const int64_t j = jb + q;
            if (j < nj) {  // wave-uniform
                double *out = ef + ((i - ef_i0) * nj + j) * row_doubles + seg0;
                const double *tq = t + q * (EF_COLS * 3);
                // Written once and read by nobody on the device: non-temporal stores (measured -10 %),
                // 16 bytes per lane and laid over the 128-byte lines of the output: the doubles in front
                // of the segment's first line boundary go out as one short store, everything behind it
                // as stores that each cover eight whole lines (measured -5 % against 8-byte stores that
                // start at the segment's first double, whatever its alignment).
                const int head = (int)(((128 - (reinterpret_cast<uintptr_t>(out) & 127)) & 127) >> 3);
                if (lane < head && lane < seg_len) __builtin_nontemporal_store(tq[lane], out + lane);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int e = head + 2 * (lane + 64 * c);
                    if (e + 1 < seg_len) {
                        const v2d v = {tq[e], tq[e + 1]};
                        __builtin_nontemporal_store(v, reinterpret_cast<v2d *>(out + e));
                    } else if (e < seg_len) {
                        __builtin_nontemporal_store(tq[e], out + e);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        above = above_next;
        below = below_next;
#pragma unroll
        for (int q = 0; q < JR; q++) {
            edge[q] = edge_next[q];
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
    // measured at 513^3: 2 rows per lane (four waves per SIMD instead of two) 7 % slower than 4 (more
    // halo rows per owned row); 8 do not fit the register file at two waves per SIMD
    constexpr int JR = 4;
    const int rows = 4 * JR;
    // columns left after the last full window are taken node by node (extra workgroups) when they are few
    const int64_t rem = n[2] % EF_COLS;
    const int64_t kmain = (n[2] > EF_COLS && rem != 0 && rem <= 4) ? n[2] - rem : n[2];
    EfGeom g;
    g.ktiles = (int)((kmain + EF_COLS - 1) / EF_COLS);
    g.jtiles = (int)((n[1] + rows - 1) / rows);
    const int64_t planes = r.hi - r.lo;
    const bool blocked = wg_order_blocked();
    g.chunks = pick_chunks((int64_t)g.ktiles * g.jtiles, planes, 3, 1, 512, !blocked);
    g.ci = (planes + g.chunks - 1) / g.chunks;
    // one more workgroup per plane, after the marching ones, when columns are left over
    const unsigned col_wgs = kmain < n[2] ? (unsigned)planes : 0u;
    const WgOrder order{g.ktiles * g.jtiles * g.chunks, blocked ? 1 : 0};
    const Coef hx = make_coef(0, 0, 0, 2 * dh[0]), hy = make_coef(0, 0, 0, 2 * dh[1]),
               hz = make_coef(0, 0, 0, 2 * dh[2]);
    hipLaunchKernelGGL((k_compute_ef<JR>), dim3(order.grid() + col_wgs), dim3(256), 0, s, phi, ef, ef_i0, n[0], kmain, hx, hy,
                       hz, r, g, order);
}

}  // namespace mgk
