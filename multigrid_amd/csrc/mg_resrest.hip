// Fused fine-grid residual + sum of squares + full-weighting restriction.
//
// Unfused, the cycle writes R_h (8 B/pt), reads it back for the norm and again for
// the restriction.  R_h has no other consumer, so here it never reaches memory: a
// workgroup marches a (j,k) window along i, computes the residual plane by plane
// into a three-plane LDS ring, adds the squares of the points it owns to its
// partial sum, and after every odd plane emits one plane of the restricted
// residual from the ring.  Traffic: phi 8 + rhs 8 + R_2h 1 = 17 bytes per fine point.
//
// Ownership: a lane owns PAIRS of adjacent k in Q vertically adjacent rows, so
// the j+-1 neighbours of a point are in the lane's own registers (except at the
// ends of its row block: two extra pair loads, cache hits), the k+-1 neighbours
// come from the adjacent lane by wave shuffle, and i+-1 are the same columns one
// plane earlier/later, also registers.  LDS holds only residual planes.
//
// Arithmetic is the reference's, term by term (cpp/PotentialSolver.cpp:3792-3839):
// the 27 restriction terms are added left to right with the first index fastest.
#include "mg_kernels.h"

#include <type_traits>


namespace mgk {

namespace {

struct alignas(16) d2 { double x, y; };

// Value of lane - 1 / lane + 1 of the wave.  Lanes 0 / 63, which have no such neighbour, get 0 (a
// shift that kept their own value would cost a register copy per half): they sit at a window-edge
// column, as do the lanes at the ends of a 32-lane row, which receive another row's value -- nothing
// computed from either is ever used.
__device__ __forceinline__ double lane_left(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_right(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// WJ x 2*WKP window of fine points; residual valid on the window minus one ring;
// owned (norm, restriction centres) = window minus two rings.
//
// GS: the convergence test of the plain Gauss-Seidel/SOR solver instead (reference
// GaussSeidelSolver::solve, cpp/PotentialSolver.cpp:267-286) -- the same march over phi and the
// right-hand side, with
//   R = -phi*d + rho/EPS_0 + cx*(phi[i-1]+phi[i+1]) + cy*(phi[j-1]+phi[j+1]) + cz*(phi[k-1]+phi[k+1])
// evaluated left to right (opposite sign and another association than the multigrid residual),
// squares summed over the INTERIOR nodes only, and nothing else: no residual planes in LDS, no
// restriction, no barrier in the plane loop.
template <int WJ, int WKP, int D, bool STORE_RF, int NT, bool GS = false>
__global__ __launch_bounds__(NT) void k_residual_restrict(View phi, View b, View Rf, View Rc, int nI,
                                                           Coef c, double *__restrict__ partials,
                                                           int CI, int nchunks, int tiles_k, int rlo,
                                                           int rhi, WgOrder order, int cstride)
{
    static_assert(WKP == 32, "row shuffles assume 32 lanes per row");
    constexpr int WK = 2 * WKP;
    constexpr int G = NT / WKP;    // row blocks per workgroup
    constexpr int Q = WJ / G;      // adjacent rows per lane
    constexpr int TJ = WJ - 4, TK = WK - 4;   // owned fine rows / columns (even)
    constexpr int TJC = TJ / 2, TKC = TK / 2; // coarse outputs per tile and coarse plane
    __shared__ double ring[GS ? 1 : 3][GS ? 1 : WJ][WK];
    __shared__ double red[NT / 64];

    const int tid = threadIdx.x;
    const int m = tid % WKP, g = tid / WKP;
    const int item = order.item(blockIdx.x);
    if (item < 0) return;  // grid padding; uniform per workgroup
    const int ntiles = order.total / nchunks;
    const int chunk = order.blocked ? item / ntiles : item % nchunks;
    const int tile = order.blocked ? item % ntiles : item / nchunks;
    const int tk = tile % tiles_k, tj = tile / tiles_k;
    const int j0 = tj * TJ - 2, k0 = tk * TK - 2;
    const int nJ = (int)phi.d[1], nK = (int)phi.d[2];
    const int cI = nI >> 1, cJ = (int)Rc.d[1], cK = (int)Rc.d[2];
    // chunks are CI planes long and start cstride planes apart (cstride == CI: one contiguous range)
    const int ia = rlo + chunk * cstride;
    const int ib = min(ia + CI, rhi);
    double acc = 0.0;
    if (ia < ib) {
        const int kk = k0 + 2 * m;
        const int kkc = min(max(kk, 0), (nK - 1) & ~1);
        // row index rq = -1..Q: the lane's Q rows plus the row above and below its block
        auto row_of = [&](int rq) { return g * Q + rq; };
        // loads clamp to the planes each view stores (views may be slabs with different halos)
        auto load = [&](const View &v, int i, int rq) -> d2 {
            const int j = min(max(j0 + row_of(rq), 0), nJ - 1);
            const int ic = min(max(i, (int)v.i0), (int)v.i_end() - 1);
            return *reinterpret_cast<const d2 *>(v.p + v.at(ic, j, kkc));
        };

        // Own rows: planes p-1, p, p+1 and D planes in flight.  The rows above and below the block
        // (hN, hS) and the right-hand side are needed at plane p only, but they too are requested
        // D planes ahead: loads return in order, so waiting for a value requested one plane ago
        // would also wait for everything requested before it and shorten the prefetch to one plane.
        d2 um1[Q], u0[Q], up1[Q], uf[D][Q], hN0, hS0, hNf[D], hSf[D], b0[Q], bf[D][Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            um1[q] = load(phi, ia - 2, q);
            u0[q] = load(phi, ia - 1, q);
            up1[q] = load(phi, ia, q);
            b0[q] = load(b, ia - 1, q);
            // the step for plane ia - 1 + d takes its requests from buffer (d + 1) % D: see the plane loop
#pragma unroll
            for (int d = 0; d + 1 < D; d++) {
                uf[(d + 1) % D][q] = load(phi, ia + 1 + d, q);
                bf[(d + 1) % D][q] = load(b, ia + d, q);
            }
        }
        hN0 = load(phi, ia - 1, -1);
        hS0 = load(phi, ia - 1, Q);
#pragma unroll
        for (int d = 0; d + 1 < D; d++) {
            hNf[(d + 1) % D] = load(phi, ia + d, -1);
            hSf[(d + 1) % D] = load(phi, ia + d, Q);
        }

        // One plane step.  The planes requested ahead land in D buffers used in turn, and the loop
        // is unrolled D times so that each step names its buffers statically (K takes this step's
        // requests, (K + 1) % D holds what this step hands on): a register with a load in flight is
        // not touched before the step that needs it, where a chain of register copies would make
        // every copy wait for its load in the step that issued it.
        auto step = [&](int p, auto kc) {
            constexpr int K = decltype(kc)::value, KU = (K + 1) % D;
            const int pl = min(p + 1 + D, ib);
            const int pb = min(p + D, ib);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                uf[K][q] = load(phi, pl, q);
                bf[K][q] = load(b, pb, q);
            }
            hNf[K] = load(phi, pb, -1);
            hSf[K] = load(phi, pb, Q);

            if constexpr (GS) {
                if (p >= ia && p >= 1 && p <= nI - 2) {
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const int r = row_of(q);
                        const int j = j0 + r;
                        const d2 u = u0[q];
                        const d2 un = q == 0 ? hN0 : u0[q > 0 ? q - 1 : 0];
                        const d2 us = q == Q - 1 ? hS0 : u0[q < Q - 1 ? q + 1 : Q - 1];
                        const double left = lane_left(u.y);
                        const double right = lane_right(u.x);
                        const double rx = -u.x * c.d + b0[q].x + c.cx * (um1[q].x + up1[q].x) + c.cy * (un.x + us.x) +
                                          c.cz * (left + u.y);
                        const double ry = -u.y * c.d + b0[q].y + c.cx * (um1[q].y + up1[q].y) + c.cy * (un.y + us.y) +
                                          c.cz * (u.x + right);
                        const bool own = r >= 2 && r <= WJ - 3 && m >= 1 && m <= WKP - 2 && j >= 1 && j <= nJ - 2;
                        if (own && kk >= 1 && kk <= nK - 2) acc += rx * rx;
                        if (own && kk + 1 <= nK - 2) acc += ry * ry;
                    }
                }
            } else if (p >= 0) {
                double(*S)[WK] = ring[p % 3];
                const bool iface = p == 0 || p == nI - 1;
                const bool iown = p >= ia;
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int r = row_of(q);
                    const int j = j0 + r;
                    const d2 u = u0[q];
                    const d2 un = q == 0 ? hN0 : u0[q > 0 ? q - 1 : 0];
                    const d2 us = q == Q - 1 ? hS0 : u0[q < Q - 1 ? q + 1 : Q - 1];
                    // k-1 / k+1 across the pair boundary: the neighbouring lane's element (at the ends of
                    // a 32-lane row the value comes from another row: window-edge columns, never used)
                    const double left = lane_left(u.y);
                    const double right = lane_right(u.x);
                    d2 rr;
                    rr.x = u.x * c.d - b0[q].x - c.cx * (um1[q].x + up1[q].x) - c.cy * (un.x + us.x) -
                           c.cz * (left + u.y);
                    rr.y = u.y * c.d - b0[q].y - c.cx * (um1[q].y + up1[q].y) - c.cy * (un.y + us.y) -
                           c.cz * (u.x + right);
                    const bool jface = j == 0 || j == nJ - 1;
                    if (iface || jface || kk == 0 || kk == nK - 1) rr.x = u.x - 0;
                    if (iface || jface || kk + 1 == 0 || kk + 1 == nK - 1) rr.y = u.y - 0;
                    *reinterpret_cast<d2 *>(&S[r][2 * m]) = rr;
                    const bool own = iown && r >= 2 && r <= WJ - 3 && m >= 1 && m <= WKP - 2 && j < nJ;
                    if (own && kk < nK) {
                        acc += rr.x * rr.x;
                        if (STORE_RF) Rf.p[Rf.at(p, j, kk)] = rr.x;
                    }
                    if (own && kk + 1 < nK) {
                        acc += rr.y * rr.y;
                        if (STORE_RF) Rf.p[Rf.at(p, j, kk + 1)] = rr.y;
                    }
                }
            }
            if constexpr (!GS) __syncthreads();
            // ---- one plane of R_2h from residual planes p-2, p-1, p ---------
            if (!GS && (p & 1) && p >= ia + 1) {
                const int I = (p - 1) >> 1;
                if (I < cI) {
                    double(*A)[WK] = ring[(p - 2 + 3) % 3];  // plane 2I-1
                    double(*B)[WK] = ring[(p - 1) % 3];      // plane 2I
                    double(*C)[WK] = ring[p % 3];            // plane 2I+1
                    // 32 lanes per coarse row: lane l holds the fine pair (2l, 2l+1) of each of the nine
                    // (di, dj) rows it needs -- one conflict-free 16-byte LDS read each -- and takes the
                    // column to its left (2l-1) from lane l-1's pair by a wave shift.  Lanes 1..TKC
                    // produce coarse column l-1; lanes 0 and 31 only feed their neighbours.  The terms
                    // are added in the reference's order: dk = -1 (left column), then 0, then +1.
                    constexpr int RPR = NT / WKP;                       // coarse rows per round
                    constexpr int ROUNDS = (TJC + RPR - 1) / RPR;
                    const int l = tid % WKP;
                    const int Kl = l - 1;
                    const int K = tk * TKC + Kl;
#pragma unroll
                    for (int o = 0; o < ROUNDS; o++) {
                        const int Jl = min(tid / WKP + o * RPR, TJC - 1);   // rows past the end repeat the last one
                        const bool rowlive = tid / WKP + o * RPR < TJC;
                        const int J = tj * TJC + Jl;
                        const int r = 2 * Jl + 2;
#define PAIR(di, dj) (*reinterpret_cast<const d2 *>(&((di) < 0 ? A : (di) == 0 ? B : C)[r + (dj)][2 * l]))
                        // (a term w * x with w = 2, 4, 8 is exact, so fma(w, x, v) rounds once, like v + w * x)
                        double v;
                        // dk = -1: the odd column left of the pair
                        v = lane_left(PAIR(-1, -1).y);
                        v = __builtin_fma(2.0, lane_left(PAIR(0, -1).y), v);
                        v = v + lane_left(PAIR(1, -1).y);
                        v = __builtin_fma(2.0, lane_left(PAIR(-1, 0).y), v);
                        v = __builtin_fma(4.0, lane_left(PAIR(0, 0).y), v);
                        v = __builtin_fma(2.0, lane_left(PAIR(1, 0).y), v);
                        v = v + lane_left(PAIR(-1, 1).y);
                        v = __builtin_fma(2.0, lane_left(PAIR(0, 1).y), v);
                        v = v + lane_left(PAIR(1, 1).y);
                        // dk = 0 (pair.x) while keeping pair.y for dk = +1
                        const d2 q0 = PAIR(-1, -1), q1 = PAIR(0, -1), q2 = PAIR(1, -1), q3 = PAIR(-1, 0), q4 = PAIR(0, 0),
                                 q5 = PAIR(1, 0), q6 = PAIR(-1, 1), q7 = PAIR(0, 1), q8 = PAIR(1, 1);
                        v = __builtin_fma(2.0, q0.x, v);
                        v = __builtin_fma(4.0, q1.x, v);
                        v = __builtin_fma(2.0, q2.x, v);
                        v = __builtin_fma(4.0, q3.x, v);
                        v = __builtin_fma(8.0, q4.x, v);
                        v = __builtin_fma(4.0, q5.x, v);
                        v = __builtin_fma(2.0, q6.x, v);
                        v = __builtin_fma(4.0, q7.x, v);
                        v = __builtin_fma(2.0, q8.x, v);
                        // dk = +1
                        v = v + q0.y;
                        v = __builtin_fma(2.0, q1.y, v);
                        v = v + q2.y;
                        v = __builtin_fma(2.0, q3.y, v);
                        v = __builtin_fma(4.0, q4.y, v);
                        v = __builtin_fma(2.0, q5.y, v);
                        v = v + q6.y;
                        v = __builtin_fma(2.0, q7.y, v);
                        v = v + q8.y;
#undef PAIR
                        v = v * 0.015625;  // == v / 64.0 exactly (power of two)
                        if (I == 0 || J == 0 || K == 0) v = q4.x;  // injection: R_2h = R_h at (2I, 2J, 2K)
                        if (rowlive && Kl >= 0 && Kl < TKC && J < cJ && K < cK) Rc.p[Rc.at(I, J, K)] = v;
                    }
                }
                __syncthreads();  // ring slot (p-2)%3 is overwritten by plane p+1 next
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                um1[q] = u0[q];
                u0[q] = up1[q];
                up1[q] = uf[KU][q];
                b0[q] = bf[KU][q];
            }
            hN0 = hNf[KU];
            hS0 = hSf[KU];
        };
        static_assert(D >= 1 && D <= 3, "the plane loop is unrolled for up to three buffers");
        for (int p = ia - 1; p < ib; p += D) {
            step(p, std::integral_constant<int, 0>{});
            if constexpr (D > 1)
                if (p + 1 < ib) step(p + 1, std::integral_constant<int, 1 % D>{});
            if constexpr (D > 2)
                if (p + 2 < ib) step(p + 2, std::integral_constant<int, 2 % D>{});
        }
    }
    // workgroup partial sum, fixed order
    acc = wave_sum64(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) t += red[w];
        partials[item] = t;
    }
}

// window rows: 32 (256 threads) or 64 (512 threads; fewer halo rows re-read per owned row, pays
// once a plane has enough windows to fill the chip).
int window_rows(const int64_t n[3]) { return n[1] >= 256 ? 64 : 32; }

struct Geometry {
    int tiles_j, tiles_k, nchunks, CI, cstride, rhi;
    WgOrder order;
    unsigned grid;  // = order.total: workgroups that do work, and partial sums written
};

// r2 non-empty: two short ranges of equal length in one launch (the planes next to either slab
// neighbour), one chunk each.
Geometry geometry(const int64_t n[3], IRange r, IRange r2 = IRange{0, 0})
{
    const int TJ = window_rows(n) - 4, TK = 60;
    Geometry g;
    const int64_t planes = r.hi - r.lo;
    g.tiles_j = (int)((n[1] + TJ - 1) / TJ);
    g.tiles_k = (int)((n[2] + TK - 1) / TK);
    const bool blocked = wg_order_blocked();
    if (r2.hi > r2.lo) {
        g.nchunks = 2;
        g.CI = (int)planes;
        g.cstride = (int)(r2.lo - r.lo);
        g.rhi = (int)r2.hi;
    } else {
        int nchunks = pick_chunks((int64_t)g.tiles_j * g.tiles_k, planes, 3, 2, window_rows(n) == 64 ? 256 : 512,
                                  !blocked);
        static const int forced = exp_knob("MG_EXP_CHUNKS_RES", 0);
        if (forced > 0 && planes >= 100) nchunks = forced;
        int CI = (int)((planes + nchunks - 1) / nchunks);
        CI += CI & 1;  // chunks start on even planes: a coarse plane's three fine planes stay together
        g.nchunks = nchunks;
        g.CI = CI;
        g.cstride = CI;
        g.rhi = (int)r.hi;
    }
    g.grid = (unsigned)(g.nchunks * g.tiles_j * g.tiles_k);
    g.order = WgOrder{(int)g.grid, blocked ? 1 : 0};
    return g;
}

}  // namespace

int64_t residual_restrict_partials(const int64_t n[3], IRange r, IRange r2) { return geometry(n, r, r2).grid; }

// Coarse planes of R_c = restrict27(residual(phi, b)) for the fine planes r (r.lo even), and
// partials[0..residual_restrict_partials) = per-workgroup sums of squares of the fine residual
// over planes r.  Rf.p may be null (the fine residual is then never written).
void residual_restrict(hipStream_t s, View phi, View b, View Rf, View Rc, const int64_t n[3], Coef c,
                       double *partials, IRange r, IRange r2)
{
    const Geometry g = geometry(n, r, r2);
#define LAUNCHW(WJ, D, RF, NT)                                                                        \
    hipLaunchKernelGGL((k_residual_restrict<WJ, 32, D, RF, NT>), dim3(g.order.grid()), dim3(NT), 0, s, phi, b, \
                       Rf, Rc, (int)n[0], c, partials, g.CI, g.nchunks, g.tiles_k, (int)r.lo, g.rhi, g.order, g.cstride)
    // Measured and not built any more: prefetch depths 1 and 3, 512 threads on the 32-row window, and
    // 1024 threads (two rows per lane) on the 64-row window (+10 %).
    if (window_rows(n) == 64) {
        if (Rf.p) LAUNCHW(64, 2, true, 512);
        else LAUNCHW(64, 2, false, 512);
    } else {
        if (Rf.p) LAUNCHW(32, 2, true, 256);
        else LAUNCHW(32, 2, false, 256);
    }
#undef LAUNCHW
}

// Convergence test of the Gauss-Seidel solver: partials[0..gs_residual_partials) = per-workgroup
// sums of R^2 over the interior nodes of planes r (see k_residual_restrict, GS).
int64_t gs_residual_partials(const int64_t n[3], IRange r) { return r.hi > r.lo ? geometry(n, r).grid : 0; }

void gs_residual_sumsq(hipStream_t s, View phi, View b, const int64_t n[3], Coef c, double *partials, IRange r)
{
    if (r.hi <= r.lo) return;
    const Geometry g = geometry(n, r);
    View none = phi;
    none.p = nullptr;
#define LAUNCHG(WJ, NT)                                                                                          \
    hipLaunchKernelGGL((k_residual_restrict<WJ, 32, 2, false, NT, true>), dim3(g.order.grid()), dim3(NT), 0, s, phi, \
                       b, none, none, (int)n[0], c, partials, g.CI, g.nchunks, g.tiles_k, (int)r.lo, g.rhi, g.order, g.cstride)
    if (window_rows(n) == 64) LAUNCHG(64, 512);
    else LAUNCHG(32, 256);
#undef LAUNCHG
}

}  // namespace mgk
