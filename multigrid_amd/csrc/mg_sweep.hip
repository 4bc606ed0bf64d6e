// Fused red-black GS-SOR sweep: both colours in ONE pass over the arrays.
//
// A colour-per-launch smoother reads and writes every cache line of u twice per
// sweep (each line holds both colours).  Here a workgroup owns a (j,k) window
// and marches along i, doing the red update of plane p and then the black
// update of plane p-1, which by then has every red neighbour it needs
// (planes p-2, p-1, p).  u is read once, the right-hand side once, u written
// once: 24 bytes per point per full sweep.
//
// The sweep is out of place (u_in -> u_out): windows overlap by two halo
// rings, and an in-place march would let one workgroup read planes a
// neighbour has already overwritten.
//
// Lane mapping: a lane owns PAIRS of adjacent k (one red, one black point), so
// every lane has work in both colour phases and all global accesses are
// 16-byte, 16-byte-aligned.  The i-1/i+1 neighbours of a point live in the
// owning lane's registers; the in-plane neighbours come from LDS copies of
// planes p and p-1.
//
// Results are bit-identical to the two-pass red-black sweep: every point sees
// exactly the operand values of the sequential colour order, and the update
// expression is the same (no contraction, true division).
#include "mg_kernels.h"
#include <type_traits>


namespace mgk {

namespace {

struct alignas(16) d2 { double x, y; };

__device__ __forceinline__ double pick(d2 v, int e) { return e ? v.y : v.x; }
__device__ __forceinline__ d2 put(d2 v, int e, double x)
{
    d2 r;
    r.x = e ? v.x : x;
    r.y = e ? x : v.y;
    return r;
}

// What to do with u_in as it is loaded, before the sweep sees it.  The prolongation that
// precedes the first sweep of a level (and the correction phi -= P(eps) that precedes the first
// fine sweep of the next iteration) only needs the coarse array, so applying it on the fly saves
// writing the prolongated field and reading it back (16 bytes per point).
enum : int {
    kLoadPlain = 0,
    kLoadCorrect = 1,  // u := u - P(ec) on 1..n-2          (reference cpp/PotentialSolver.cpp:4170-4214)
    kLoadProlong = 2,  // u := P(ec) on 0..n-2, 0 at index 0 (reference cpp/PotentialSolver.cpp:3940-3976)
    // The same for a u_in known to be zero outside 0..n-2 per axis, which is then not read at all
    // (17 instead of 25 bytes per point).  True of every eps_l inside the solver: the prolongation
    // writes 0..n_l-2, the smoother 1..n_l-2, and nothing else ever writes the array.
    kLoadProlongZero = 3
};
constexpr bool is_prolong(int mode) { return mode == kLoadProlong || mode == kLoadProlongZero; }


// One colour phase of the sweep for the Q points a lane owns in a plane.  E (which element of
// the lane's pairs has this colour) is a template argument: the caller branches on it once per
// phase with a wave-uniform condition, so no per-point selects remain.  P points at the lane's
// (row rw, pair m) element of the plane's LDS slot; in-plane neighbours are constant offsets.
template <int E> __device__ __forceinline__ double &elem(d2 &v) { return E ? v.y : v.x; }
template <int E> __device__ __forceinline__ double elem(const d2 &v) { return E ? v.y : v.x; }

// Value of lane - 1 / lane + 1 of the wave (lanes 0 / 63, which have no such neighbour, get 0).
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

// One colour phase on a plane whose LDS slot keeps the two pair elements apart: element e of
// (row r, pair m) at e * HALF + (r + 1) * WKP + m.  The j-1 / j+1 neighbours of a point are the
// same element one row up / down (consecutive lanes read consecutive words: no bank conflicts);
// of the k-1 / k+1 neighbours one is the lane's other pair element and one the adjacent lane's,
// fetched by a wave shift (at the ends of a 32-lane row it comes from another row: window-edge
// columns, never used).  P points at element 0 of (row rw - 1, pair m), the row above the
// lane's first one, so every access is P plus a non-negative constant; vc is the plane being
// updated, va / vb its i-1 / i+1 neighbours, all as the lane's registers hold them.
// okmask: bit E * Q + q = the point may be updated.  PUBLISH: new values also go to the slot.
template <int E, int Q, int ROWSTEP, int WKP, int HALF, bool PUBLISH, bool FAST>
__device__ __forceinline__ void phase_split(double *P, d2 (&vc)[Q], const d2 (&va)[Q], const d2 (&vb)[Q],
                                            const d2 (&b)[Q], unsigned okmask, const Coef &c, double omega)
{
    // (the two forms of the update -- see sor_n -- are spelled out here: the sums go straight into
    // the expression, which keeps a lane's four rows within the register budget)
    if constexpr (FAST) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const double *X = P + E * HALF + q * ROWSTEP;
            const double mate = elem<1 - E>(vc[q]);
            const double far = E ? lane_right(mate) : lane_left(mate);
            const double uc = elem<E>(vc[q]);
            double t = fma(c.fb, elem<E>(b[q]), c.fu * uc);
            t = fma(c.fx, elem<E>(va[q]) + elem<E>(vb[q]), t);
            t = fma(c.fy, X[0] + X[2 * WKP], t);
            t = fma(c.fz, E ? mate + far : far + mate, t);
            const bool ok = (okmask >> (E * Q + q)) & 1u;
            const double un = ok ? t : uc;
            elem<E>(vc[q]) = un;
            if (PUBLISH) P[E * HALF + q * ROWSTEP + WKP] = un;
        }
    } else {
        double x[Q], g[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const double *X = P + E * HALF + q * ROWSTEP;
            const double mate = elem<1 - E>(vc[q]);
            const double far = E ? lane_right(mate) : lane_left(mate);
            x[q] = elem<E>(b[q]) + c.cx * (elem<E>(va[q]) + elem<E>(vb[q])) + c.cy * (X[0] + X[2 * WKP]) +
                   c.cz * (E ? mate + far : far + mate);
        }
        div_d_n<Q>(x, c, g);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const double uc = elem<E>(vc[q]);
            const bool ok = (okmask >> (E * Q + q)) & 1u;
            const double un = ok ? uc + omega * (g[q] - uc) : uc;
            elem<E>(vc[q]) = un;
            if (PUBLISH) P[E * HALF + q * ROWSTEP + WKP] = un;
        }
    }
}

// WJ rows x 2*WKP columns per window; outputs are the window minus two rings.
// D = planes of u (and of the right-hand side) requested ahead of use: the march is
// latency bound unless several planes per workgroup are in flight.
template <int WJ, int WKP, int D, int NT, int MODE, bool FAST>
__global__ __launch_bounds__(NT) void k_sweep_rb(View uin, View uout, View rhs, View ec, int nI, int nJ,
                                                  int nK, Coef c, double omega, int CI, int nchunks,
                                                  int tiles_k, int rlo, int rhi, WgOrder order, int cstride)
{
    constexpr int WK = 2 * WKP;
    constexpr int TJ = WJ - 4, TK = WK - 4;
    constexpr int RPP = NT / WKP;   // rows covered by one pass of the workgroup
    constexpr int Q = WJ / RPP;     // pair-columns owned by a thread
    constexpr int CJ = WJ / 2 + 1, CK = WKP + 1;  // coarse footprint of the window
    constexpr int CKP = CK + 1;                   // padded row
    constexpr int CE = (CJ * CK + NT - 1) / NT;   // coarse elements staged per thread
    static_assert(WJ % RPP == 0, "window rows must be a multiple of rows per pass");
    static_assert(RPP % 4 == 0 && TJ % 2 == 0, "row parity must be the same for all rows of a wave");
    // Three plane slots, each with a guard row above and below the window: the in-plane
    // neighbours of any window point are then plain constant offsets from the point (the values
    // read across a window edge are garbage that only reaches points which are not updated).
    static_assert(WKP == 32, "the wave shifts assume 32 lanes per row");
    constexpr int HALF = (WJ + 2) * WKP;  // one pair element of a plane (see phase_split)
    constexpr int PL = 2 * HALF;          // doubles per slot
    constexpr int RS = RPP * WKP;         // from owned row q to q + 1
    __shared__ double lds[3 * PL];
    __shared__ double cl[2][CJ][CKP];  // staged coarse planes; unused (and not allocated) in plain mode

    const int tid = threadIdx.x;
    const int m = tid % WKP, rr = tid / WKP;
    // Row of this lane within a pass of RPP rows.  The two 32-lane halves of a wave take rows
    // RPP/2 apart instead of adjacent ones: rows of equal parity have their red point in the same
    // pair element and the same prolongation case, so the wave does not diverge.
    const int rw = (rr >> 1) + (RPP / 2) * (rr & 1);
    // element 0 of (row rw - 1, pair m) inside a slot: what phase_split addresses from
    const int tbase = rw * WKP + m;
    // plain order: chunk index fastest (with a multiple of 8 chunks all windows of a chunk share
    // blockIdx % 8, i.e. one XCD); blocked order: chunk slowest, each XCD a contiguous run of windows
    const int item = order.item(blockIdx.x);
    if (item < 0) return;  // grid padding; uniform per workgroup
    const int ntiles = order.total / nchunks;
    const int chunk = order.blocked ? item / ntiles : item % nchunks;
    const int tile = order.blocked ? item % ntiles : item / nchunks;
    const int tk = tile % tiles_k, tj = tile / tiles_k;
    const int j0 = tj * TJ - 2, k0 = tk * TK - 2;
    const int uJ = (int)uin.d[1], uK = (int)uin.d[2];
    const int rJ = (int)rhs.d[1], rK = (int)rhs.d[2];
    // stored plane ranges of the (possibly slab) views, global indices
    const int ulo = (int)uin.i0, uhi = (int)uin.i_end() - 1;
    const int blo = (int)rhs.i0, bhi = (int)rhs.i_end() - 1;
    // chunks are CI planes long and start cstride planes apart (cstride == CI: one contiguous range)
    const int ia = rlo + chunk * cstride;
    const int ib = min(ia + CI, rhi);
    if (ia >= ib) return;  // uniform per workgroup
    const int kk = k0 + 2 * m;  // first column of this lane's pairs

    // Loads are unconditional with clamped coordinates: a branch around a load makes
    // the compiler wait for it on the spot, which serialises the whole prefetch.
    // Clamped (out-of-array) lanes fetch some valid element that is never used.
    const int kku = min(max(kk, 0), (uK - 1) & ~1);
    const int kkr = min(max(kk, 0), (rK - 1) & ~1);
    auto load_u = [&](int i, int q) -> d2 {
        if constexpr (MODE == kLoadProlongZero) return d2{0.0, 0.0};  // zero wherever fix() keeps it
        const int j = min(max(j0 + q * RPP + rw, 0), uJ - 1);
        const int ic = min(max(i, ulo), uhi);
        return *reinterpret_cast<const d2 *>(uin.p + uin.at(ic, j, kku));
    };
    auto load_b = [&](int i, int q) -> d2 {
        const int j = min(max(j0 + q * RPP + rw, 0), rJ - 1);
        const int ic = min(max(i, blo), bhi);
        return *reinterpret_cast<const d2 *>(rhs.p + rhs.at(ic, j, kkr));
    };

    // ---- coarse tile staging (MODE != plain): coarse planes I and I+1 live in cl[I&1], cl[(I+1)&1]
    const int J0 = j0 >> 1, K0 = k0 >> 1;  // j0, k0 are even (arithmetic shift keeps -2 -> -1)
    auto coarse_elem = [&](int I, int e) -> double {
        const int Jl = e / CK, Kl = e % CK;
        const int Ic = min(max(I, (int)ec.i0), (int)ec.i_end() - 1);
        const int Jc = min(max(J0 + Jl, 0), (int)ec.d[1] - 1);
        const int Kc = min(max(K0 + Kl, 0), (int)ec.d[2] - 1);
        return ec.p[ec.at(Ic, Jc, Kc)];
    };
    auto stage_write = [&](int I, const double (&v)[CE]) {
#pragma unroll
        for (int t = 0; t < CE; t++) {
            const int e = tid + t * NT;
            if (e < CJ * CK) cl[I & 1][e / CK][e % CK] = v[t];
        }
    };
    // Loop-invariant part of the on-load operation's range tests, one bit per owned row q:
    // correct: bit q = x may be corrected (j in 1..nJ-2, k in 1..nK-2), bit Q+q = y may;
    // prolong: bit q / Q+q = x / y lies in the loop range (j <= nJ-2, k <= nK-2),
    //          bit 2Q+q / 3Q+q = x / y is forced to zero there (j == 0 or k == 0).
    unsigned fmask = 0;
    if (MODE != kLoadPlain) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = j0 + q * RPP + rw;
            if (MODE == kLoadCorrect) {
                const bool jin = j >= 1 && j <= nJ - 2;
                fmask |= (jin && kk >= 1 && kk <= nK - 2 ? 1u : 0u) << q;
                fmask |= (jin && kk + 1 <= nK - 2 ? 1u : 0u) << (Q + q);
            } else {
                const bool jin = j <= nJ - 2;
                fmask |= (jin && kk <= nK - 2 ? 1u : 0u) << q;
                fmask |= (jin && kk + 1 <= nK - 2 ? 1u : 0u) << (Q + q);
                fmask |= (j == 0 || kk == 0 ? 1u : 0u) << (2 * Q + q);
                fmask |= (j == 0 ? 1u : 0u) << (3 * Q + q);
            }
        }
    }
    // every row of a wave has the parity of rw (see the colour phases below)
    const bool oj = __builtin_amdgcn_readfirstlane((j0 + rw) & 1) != 0;
    // u as the sweep must see it: the loaded pair of (plane f, owned row q) after the on-load
    // operation.  Reads the staged coarse planes f>>1 and (f>>1)+1.  The interpolation case
    // (parities of f and j) is wave-uniform: scalar branches.
    auto fix = [&](d2 raw, int f, int q) -> d2 {
        if constexpr (MODE == kLoadPlain) return raw;
        const bool fin = f <= nI - 2 && (MODE != kLoadCorrect || f >= 1);  // plane inside the operation's range
        const int I = f >> 1, Jl = (q * RPP + rw) >> 1, Kl = m;
        const double(*A)[CKP] = cl[I & 1];
        const double(*B)[CKP] = cl[(I + 1) & 1];
        const bool oi = f & 1;
        // the eight corner values; the sums below follow the reference's association order
        // all eight corners are read up front (independent LDS reads in flight together); the case
        // only decides which of them enter the sums
        const double c000 = A[Jl][Kl], c010 = A[Jl + 1][Kl], c001 = A[Jl][Kl + 1], c011 = A[Jl + 1][Kl + 1];
        const double c100 = B[Jl][Kl], c110 = B[Jl + 1][Kl], c101 = B[Jl][Kl + 1], c111 = B[Jl + 1][Kl + 1];
        double vx, vy;  // interpolant at k = kk (even) and kk+1 (odd)
        if (!oi && !oj) {
            vx = c000;
            vy = 0.5 * (c000 + c001);
        } else if (oi && !oj) {
            const double s = c000 + c100;
            vx = 0.5 * s;
            vy = 0.25 * ((s + c001) + c101);
        } else if (!oi && oj) {
            const double s = c000 + c010;
            vx = 0.5 * s;
            vy = 0.25 * ((s + c001) + c011);
        } else {
            const double s = ((c000 + c100) + c010) + c110;
            vx = 0.25 * s;
            vy = 0.125 * ((((s + c001) + c101) + c011) + c111);
        }
        d2 out = raw;
        const bool mx = fin && ((fmask >> q) & 1u), my = fin && ((fmask >> (Q + q)) & 1u);
        if (MODE == kLoadCorrect) {
            if (mx) out.x = raw.x - vx;
            if (my) out.y = raw.y - vy;
        } else {
            const bool zx = f == 0 || ((fmask >> (2 * Q + q)) & 1u), zy = f == 0 || ((fmask >> (3 * Q + q)) & 1u);
            if (mx) out.x = zx ? 0.0 : vx;
            if (my) out.y = zy ? 0.0 : vy;
        }
        return out;
    };

    // u planes p-2, p-1, p, p+1 of the owned pair-columns, then the D planes in flight;
    // right-hand side of planes p-1, p, then D in flight
    d2 vm2[Q], vm1[Q], v0[Q], vp1[Q], vf[D][Q], bm1[Q], b0[Q], bf[D][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        vm2[q] = d2{0.0, 0.0};
        vm1[q] = load_u(ia - 2, q);
        v0[q] = load_u(ia - 1, q);
        vp1[q] = load_u(ia, q);
        bm1[q] = d2{0.0, 0.0};
        b0[q] = load_b(ia - 1, q);
#pragma unroll
        for (int d = 0; d + 1 < D; d++) {
            vf[d][q] = load_u(ia + 1 + d, q);
            bf[d][q] = load_b(ia + d, q);
        }
    }
    if (MODE != kLoadPlain) {
        // The three planes loaded above are needed at once: fix them plane by plane, staging the
        // coarse planes each one needs (two barriers per step; once per workgroup).  Planes ia-2 ..
        // ia lie between coarse planes (ia-2)>>1 and that plus two: all three are requested here,
        // together with the loads above, so the steps below wait for memory once and not six times.
        const int I0 = (ia - 2) >> 1;
        double c0[CE], c1[CE], c2[CE];
#pragma unroll
        for (int t = 0; t < CE; t++) {
            c0[t] = coarse_elem(I0, tid + t * NT);
            c1[t] = coarse_elem(I0 + 1, tid + t * NT);
            c2[t] = coarse_elem(I0 + 2, tid + t * NT);
        }
#pragma unroll
        for (int step = 0; step < 3; step++) {
            const int f = ia - 2 + step, I = f >> 1;  // I - I0 is 0 or 1
            __syncthreads();
            if (I == I0) {
                stage_write(I, c0);
                stage_write(I + 1, c1);
            } else {
                stage_write(I, c1);
                stage_write(I + 1, c2);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if (step == 0) vm1[q] = fix(vm1[q], f, q);
                else if (step == 1) v0[q] = fix(v0[q], f, q);
                else vp1[q] = fix(vp1[q], f, q);
            }
        }
        // staged now: planes ia>>1 and (ia>>1)+1, which is also what plane ia+1 needs
    }
    // slot of plane p - back, given sp = slot of plane p (a small loop-carried index, so that the
    // slot offset folds into the LDS instructions' immediate offsets)
    int sp = (ia - 1 + 3) % 3;
    auto slot = [&](int back) -> double * {
        int t = sp - back;
        t = t < 0 ? t + 3 : (t >= 3 ? t - 3 : t);
        return lds + (unsigned)(t * PL + tbase);
    };
    auto fill = [&](int back, const d2 (&v)[Q]) {
        double *S = slot(back);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            S[WKP + q * RS] = v[q].x;
            S[WKP + HALF + q * RS] = v[q].y;
        }
    };
    fill(0, v0);

    // Which of this lane's points may be updated, per phase (red: window minus one ring, black:
    // minus two), pair element E and row q: bit (phase*2 + E)*Q + q.  Loop invariant.
    unsigned okmask = 0;
#pragma unroll
    for (int ph = 0; ph < 2; ph++)
#pragma unroll
        for (int E = 0; E < 2; E++)
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int r = q * RPP + rw, j = j0 + r, col = 2 * m + E, k = k0 + col;
                const bool ok = j >= 1 && j <= nJ - 2 && k >= 1 && k <= nK - 2 && r >= 1 + ph && r <= WJ - 2 - ph &&
                                col >= 1 + ph && col <= WK - 2 - ph;
                okmask |= (ok ? 1u : 0u) << ((ph * 2 + E) * Q + q);
            }

    for (int p = ia - 1; p <= ib; p++) {
        // plane f = p+2 is fixed at the end of this iteration.  When f is even it needs coarse
        // plane f/2 only (already staged); plane f/2+1, needed from f+1 on, is fetched now and
        // written to the slot plane f/2-1 occupied (last read for f-1, two barriers ago).
        // These loads are issued BEFORE the plane prefetch below: loads return in order, so the
        // wait for them (after the red phase) leaves the younger prefetch loads in flight.
        const int f = p + 2;
        double cv[CE];
        if (MODE != kLoadPlain && !(f & 1)) {
#pragma unroll
            for (int t = 0; t < CE; t++) cv[t] = coarse_elem((f >> 1) + 1, tid + t * NT);
        }
#pragma unroll
        for (int q = 0; q < Q; q++) {
            vf[D - 1][q] = load_u(min(p + 1 + D, ib + 1), q);
            bf[D - 1][q] = load_b(min(p + D, ib), q);
        }
        __syncthreads();
        // ---- red points of plane p (neighbours: old black values) ----------
        // Every row of this lane has the parity of rw (j0 and RPP are even), and the two halves
        // of a wave take rows of equal parity: which element of a pair is red is wave-uniform.
        const bool e1 = __builtin_amdgcn_readfirstlane((p + j0 + rw) & 1) != 0;
        if (p >= 1 && p <= nI - 2) {
            if (e1) phase_split<1, Q, RS, WKP, HALF, true, FAST>(slot(0), v0, vm1, vp1, b0, okmask, c, omega);
            else phase_split<0, Q, RS, WKP, HALF, true, FAST>(slot(0), v0, vm1, vp1, b0, okmask, c, omega);
        }
        if (MODE != kLoadPlain && !(f & 1)) stage_write((f >> 1) + 1, cv);
        // No barrier between the phases: the black update of plane p-1 reads, from other lanes,
        // only the red values of plane p-1 (written one iteration ago, before this iteration's
        // barrier) in a slot nobody writes now; its i neighbours are this lane's own registers.
        // The red values written above are first read by other lanes in the next iteration, the
        // coarse planes staged above after the next barrier too.
        // ---- black points of plane p-1 (neighbours: new red values), store --
        const int pb = p - 1;
        if (pb >= ia && pb < ib) {
            if (pb >= 1 && pb <= nI - 2) {
                // black of plane p-1 sits where red of plane p does
                if (e1) phase_split<1, Q, RS, WKP, HALF, false, FAST>(slot(1), vm1, vm2, v0, bm1, okmask >> (2 * Q), c, omega);
                else phase_split<0, Q, RS, WKP, HALF, false, FAST>(slot(1), vm1, vm2, v0, bm1, okmask >> (2 * Q), c, omega);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int r = q * RPP + rw;
                const int j = j0 + r;
                if (r >= 2 && r <= WJ - 3 && m >= 1 && m <= WKP - 2 && j < uJ && kk < uK)
                    *reinterpret_cast<d2 *>(uout.p + uout.at(pb, j, kk)) = vm1[q];
            }
        }
        // plane p+1 into the LDS slot plane p-2 occupied; plane p+2 arrives and is fixed
        fill(-1, vp1);
        sp = sp == 2 ? 0 : sp + 1;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            vm2[q] = vm1[q];
            vm1[q] = v0[q];
            v0[q] = vp1[q];
            vp1[q] = fix(vf[0][q], f, q);
            bm1[q] = b0[q];
            b0[q] = bf[0][q];
#pragma unroll
            for (int d = 0; d + 1 < D; d++) {
                vf[d][q] = vf[d + 1][q];
                bf[d][q] = bf[d + 1][q];
            }
        }
    }
}

// TWO full red-black sweeps in one pass over the arrays (12 bytes per point and sweep instead
// of 24): the second sweep follows the first two planes behind.  Iteration p of the march does
//   sweep 1: red of plane p, black of plane p-1      (as k_sweep_rb)
//   sweep 2: red of plane p-2, black of plane p-3, store plane p-3
// Plane p-1 is finished for sweep 1 after its black update, which is all the red update of plane
// p-2 in sweep 2 waits for; every point sees exactly the operand values of two sweeps run one
// after the other, so the result is bit-identical to two launches of k_sweep_rb.
//
// Five LDS slots hold planes p-3 .. p+1.  Each phase reads, from other lanes, only values
// written in an earlier iteration (one barrier per iteration) and writes only positions of the
// colour nobody reads in that phase: red positions of slot p, black of slot p-1 (read by sweep 2
// an iteration later), red of slot p-2; the black update of plane p-3 goes to registers and out.
// Outputs are the window minus four rings.
template <int WJ, int WKP, int D, int NT, bool FAST>
__global__ __launch_bounds__(NT) void k_sweep_rb2(View uin, View uout, View rhs, int nI, int nJ, int nK,
                                                   Coef c, double omega, int CI, int nchunks, int tiles_k,
                                                   int rlo, int rhi, WgOrder order)
{
    constexpr int WK = 2 * WKP;
    constexpr int TJ = WJ - 8, TK = WK - 8;
    constexpr int RPP = NT / WKP;
    constexpr int Q = WJ / RPP;
    static_assert(WJ % RPP == 0, "window rows must be a multiple of rows per pass");
    static_assert(RPP % 4 == 0 && TJ % 2 == 0, "row parity must be the same for all rows of a wave");
    static_assert(8 * Q <= 32, "update mask has 8 bits per owned row");
    static_assert(WKP == 32, "the wave shifts assume 32 lanes per row");
    constexpr int HALF = (WJ + 2) * WKP;  // one pair element of a plane, with a guard row above and below
    constexpr int PL = 2 * HALF;
    constexpr int NS = 5;
    constexpr int RS = RPP * WKP;         // from owned row q to q + 1
    __shared__ double lds[NS * PL];

    const int tid = threadIdx.x;
    const int m = tid % WKP, rr = tid / WKP;
    const int rw = (rr >> 1) + (RPP / 2) * (rr & 1);
    const int tbase = rw * WKP + m;  // element 0 of (row rw - 1, pair m) in a slot
    const int item = order.item(blockIdx.x);
    if (item < 0) return;
    const int ntiles = order.total / nchunks;
    const int chunk = order.blocked ? item / ntiles : item % nchunks;
    const int tile = order.blocked ? item % ntiles : item / nchunks;
    const int tk = tile % tiles_k, tj = tile / tiles_k;
    const int j0 = tj * TJ - 4, k0 = tk * TK - 4;
    const int uJ = (int)uin.d[1], uK = (int)uin.d[2];
    const int rJ = (int)rhs.d[1], rK = (int)rhs.d[2];
    const int ulo = (int)uin.i0, uhi = (int)uin.i_end() - 1;
    const int blo = (int)rhs.i0, bhi = (int)rhs.i_end() - 1;
    const int ia = rlo + chunk * CI;
    const int ib = min(ia + CI, rhi);
    if (ia >= ib) return;
    const int kk = k0 + 2 * m;
    const int kku = min(max(kk, 0), (uK - 1) & ~1);
    const int kkr = min(max(kk, 0), (rK - 1) & ~1);
    auto load_u = [&](int i, int q) -> d2 {
        const int j = min(max(j0 + q * RPP + rw, 0), uJ - 1);
        const int ic = min(max(i, ulo), uhi);
        return *reinterpret_cast<const d2 *>(uin.p + uin.at(ic, j, kku));
    };
    auto load_b = [&](int i, int q) -> d2 {
        const int j = min(max(j0 + q * RPP + rw, 0), rJ - 1);
        const int ic = min(max(i, blo), bhi);
        return *reinterpret_cast<const d2 *>(rhs.p + rhs.at(ic, j, kkr));
    };
    // slot of plane p - back, given sp = slot of plane p (kept as a small loop-carried index: the
    // slot offset then folds into the LDS instructions' immediate offsets)
    int sp = (ia + 2 * NS - 3) % NS;
    auto slot = [&](int back) -> double * {
        int t = sp - back;
        t = t < 0 ? t + NS : (t >= NS ? t - NS : t);
        return lds + (unsigned)(t * PL + tbase);
    };

    // u planes p-4 .. p+1 and D requested ahead; right-hand side of planes p-3 .. p and D ahead.
    // The requested planes land in D buffers used in turn (the loop is unrolled D times so that
    // each iteration names its buffers statically): a register with a load in flight is not
    // touched until the iteration that needs its plane.  Passing in-flight values down a chain of
    // register copies would make every copy wait for its load at the end of the iteration that
    // issued it, which serialises the memory latency with the arithmetic.
    d2 vm4[Q], vm3[Q], vm2[Q], vm1[Q], v0[Q], vp1[Q], vf[D][Q], bm3[Q], bm2[Q], bm1[Q], b0[Q], bf[D][Q];
    const int p0 = ia - 3;  // first plane whose red update (sweep 1) the outputs depend on
#pragma unroll
    for (int q = 0; q < Q; q++) {
        vm4[q] = vm3[q] = vm2[q] = d2{0.0, 0.0};
        bm3[q] = bm2[q] = bm1[q] = d2{0.0, 0.0};
        vm1[q] = load_u(p0 - 1, q);
        v0[q] = load_u(p0, q);
        vp1[q] = load_u(p0 + 1, q);
        b0[q] = load_b(p0, q);
        // iteration p0 + d takes plane p0 + d + 2 (right-hand side p0 + d + 1) from buffer (d + 1) % D
#pragma unroll
        for (int d = 0; d + 1 < D; d++) {
            vf[(d + 1) % D][q] = load_u(p0 + 2 + d, q);
            bf[(d + 1) % D][q] = load_b(p0 + 1 + d, q);
        }
    }
    auto fill = [&](int back, const d2 (&v)[Q]) {
        double *S = slot(back);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            S[WKP + q * RS] = v[q].x;
            S[WKP + HALF + q * RS] = v[q].y;
        }
    };
    fill(0, v0);

    // bit (ph * 2 + E) * Q + q: the point of phase ph (0: red 1, 1: black 1, 2: red 2, 3: black 2),
    // pair element E, owned row q is an interior point of the level
    unsigned okmask = 0;
#pragma unroll
    for (int ph = 0; ph < 4; ph++)
#pragma unroll
        for (int E = 0; E < 2; E++)
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int r = q * RPP + rw, j = j0 + r, col = 2 * m + E, k = k0 + col;
                const bool ok = j >= 1 && j <= nJ - 2 && k >= 1 && k <= nK - 2 && r >= 1 + ph && r <= WJ - 2 - ph &&
                                col >= 1 + ph && col <= WK - 2 - ph;
                okmask |= (ok ? 1u : 0u) << ((ph * 2 + E) * Q + q);
            }

    const int plast = ib + 2;  // black of sweep 2 reaches plane ib-1 at p = ib+2
    // One plane step.  K = (p - p0) % D: the buffer this step's requests go to; its plane p+2
    // (right-hand side p+1) waits in buffer (K + 1) % D.
    auto step = [&](int p, auto kc) {
        constexpr int K = decltype(kc)::value, KU = (K + 1) % D;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            vf[K][q] = load_u(min(p + 1 + D, plast + 1), q);
            bf[K][q] = load_b(min(p + D, plast), q);
        }
        __syncthreads();
        // which element of this lane's pairs is red in plane p (wave-uniform, see k_sweep_rb); the
        // same element is black in plane p-1, red in p-2, black in p-3
        const bool e1 = __builtin_amdgcn_readfirstlane((p + j0 + rw) & 1) != 0;
        if (p >= 1 && p <= nI - 2) {
            if (e1) phase_split<1, Q, RS, WKP, HALF, true, FAST>(slot(0), v0, vm1, vp1, b0, okmask, c, omega);
            else phase_split<0, Q, RS, WKP, HALF, true, FAST>(slot(0), v0, vm1, vp1, b0, okmask, c, omega);
        }
        const int pb = p - 1;
        if (pb >= p0 + 1 && pb >= 1 && pb <= nI - 2) {
            if (e1) phase_split<1, Q, RS, WKP, HALF, true, FAST>(slot(1), vm1, vm2, v0, bm1, okmask >> (2 * Q), c, omega);
            else phase_split<0, Q, RS, WKP, HALF, true, FAST>(slot(1), vm1, vm2, v0, bm1, okmask >> (2 * Q), c, omega);
        }
        const int pr = p - 2;
        if (pr >= p0 + 2 && pr >= 1 && pr <= nI - 2) {
            if (e1) phase_split<1, Q, RS, WKP, HALF, true, FAST>(slot(2), vm2, vm3, vm1, bm2, okmask >> (4 * Q), c, omega);
            else phase_split<0, Q, RS, WKP, HALF, true, FAST>(slot(2), vm2, vm3, vm1, bm2, okmask >> (4 * Q), c, omega);
        }
        const int pk = p - 3;
        if (pk >= ia) {
            if (pk >= 1 && pk <= nI - 2) {
                if (e1) phase_split<1, Q, RS, WKP, HALF, false, FAST>(slot(3), vm3, vm4, vm2, bm3, okmask >> (6 * Q), c, omega);
                else phase_split<0, Q, RS, WKP, HALF, false, FAST>(slot(3), vm3, vm4, vm2, bm3, okmask >> (6 * Q), c, omega);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int r = q * RPP + rw;
                const int j = j0 + r;
                if (r >= 4 && r <= WJ - 5 && m >= 2 && m <= WKP - 3 && j < uJ && kk < uK)
                    *reinterpret_cast<d2 *>(uout.p + uout.at(pk, j, kk)) = vm3[q];
            }
        }
        fill(-1, vp1);
        sp = sp + 1 == NS ? 0 : sp + 1;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            vm4[q] = vm3[q];
            vm3[q] = vm2[q];
            vm2[q] = vm1[q];
            vm1[q] = v0[q];
            v0[q] = vp1[q];
            vp1[q] = vf[KU][q];
            bm3[q] = bm2[q];
            bm2[q] = bm1[q];
            bm1[q] = b0[q];
            b0[q] = bf[KU][q];
        }
    };
    for (int p = p0; p <= plast; p += D) {
        step(p, std::integral_constant<int, 0>{});
        if constexpr (D > 1)
            if (p + 1 <= plast) step(p + 1, std::integral_constant<int, 1 % D>{});
        if constexpr (D > 2)
            if (p + 2 <= plast) step(p + 2, std::integral_constant<int, 2 % D>{});
        static_assert(D <= 3, "the plane loop is unrolled for up to three buffers");
    }
}

}  // namespace

// Two sweeps in one launch (k_sweep_rb2).  u_in must hold planes [r.lo-4, r.hi+4) as far as they
// exist in the level.
template <int WJ, int NT, int D = 2>
static void launch_sweep2_t(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega,
                            IRange r)
{
    constexpr int WKP = 32;
    constexpr int TJ = WJ - 8, TK = 2 * WKP - 8;
    const int64_t planes = r.hi - r.lo;
    if (planes <= 0) return;
    const int tiles_j = (int)((uin.d[1] + TJ - 1) / TJ);
    const int tiles_k = (int)((uin.d[2] + TK - 1) / TK);
    const int tiles = tiles_j * tiles_k;
    const int nchunks = pick_chunks(tiles, planes, 8, 1, 256, false);
    const int CI = (int)((planes + nchunks - 1) / nchunks);
    const WgOrder order{nchunks * tiles, 1};
    if (c.fast)
        hipLaunchKernelGGL((k_sweep_rb2<WJ, WKP, D, NT, true>), dim3(order.grid()), dim3(NT), 0, s, uin, uout, rhs,
                           (int)n[0], (int)n[1], (int)n[2], c, omega, CI, nchunks, tiles_k, (int)r.lo, (int)r.hi, order);
    else
        hipLaunchKernelGGL((k_sweep_rb2<WJ, WKP, D, NT, false>), dim3(order.grid()), dim3(NT), 0, s, uin, uout, rhs,
                           (int)n[0], (int)n[1], (int)n[2], c, omega, CI, nchunks, tiles_k, (int)r.lo, (int)r.hi, order);
}

void sweep_rb_fused2(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega,
                     IRange r)
{
    // Measured on levels of 129^3, 257^3 and 513^3 and not built any more: 32-row windows with 512
    // threads (+25 %) and three planes requested ahead instead of two (no change).
    launch_sweep2_t<48, 768>(s, uin, uout, rhs, n, c, omega, r);
}

template <int WJ, int NT, int D = 2>
static void launch_sweep_t(hipStream_t s, View uin, View uout, View rhs, View ec, int load_op,
                           const int64_t n[3], Coef c, double omega, IRange r, IRange r2)
{
    constexpr int WKP = 32;
    constexpr int TJ = WJ - 4, TK = 2 * WKP - 4;
    const int64_t planes = r.hi - r.lo;
    if (planes <= 0) return;
    const int tiles_j = (int)((uin.d[1] + TJ - 1) / TJ);
    const int tiles_k = (int)((uin.d[2] + TK - 1) / TK);
    const int tiles = tiles_j * tiles_k;
    const bool blocked = wg_order_blocked();
    int nchunks = pick_chunks(tiles, planes, 4, 1, WJ == 32 ? 512 : 256, !blocked);
    int CI = (int)((planes + nchunks - 1) / nchunks);
    int cstride = CI, rhi = (int)r.hi;
    if (r2.hi > r2.lo) {
        // two short ranges of equal length in one launch (the planes next to either slab
        // neighbour): one chunk each
        nchunks = 2;
        CI = (int)planes;
        cstride = (int)(r2.lo - r.lo);
        rhi = (int)r2.hi;
    }
    const WgOrder order{nchunks * tiles, blocked ? 1 : 0};
    const unsigned grid = order.grid();
#define LAUNCH_F(MODE, FAST)                                                                               \
    hipLaunchKernelGGL((k_sweep_rb<WJ, WKP, D, NT, MODE, FAST>), dim3(grid), dim3(NT), 0, s, uin, uout, rhs, \
                       ec, (int)n[0], (int)n[1], (int)n[2], c, omega, CI, nchunks, tiles_k, (int)r.lo,       \
                       rhi, order, cstride)
#define LAUNCH(MODE)                     \
    do {                                 \
        if (c.fast) LAUNCH_F(MODE, true); \
        else LAUNCH_F(MODE, false);      \
    } while (0)
    if (load_op == kLoadCorrect) LAUNCH(kLoadCorrect);
    else if (load_op == kLoadProlong) LAUNCH(kLoadProlong);
    else if (load_op == kLoadProlongZero) LAUNCH(kLoadProlongZero);
    else LAUNCH(kLoadPlain);
#undef LAUNCH
#undef LAUNCH_F
}

// One full red-black sweep u_in -> u_out: planes r of u_out, updated over 1..n-2 per axis
// (n = level size); every other element of those planes is copied through unchanged.
// load_op (kLoad*) is applied to u_in on the fly with the coarse array ec.
static void launch_sweep(hipStream_t s, View uin, View uout, View rhs, View ec, int load_op,
                         const int64_t n[3], Coef c, double omega, IRange r, IRange r2)
{
    // 64-row windows (512 threads) re-read fewer halo rows per owned row; they pay once a plane has
    // enough windows to fill the chip (measured: level with 257 rows the same either way, 513 rows
    // 0.69 against 0.86 ms)
    const int wj = uin.d[1] >= 256 ? 64 : 32;
    // Measured at 513^3 and not built any more: 1024 threads per window and prefetch depths 1 and 3
    // (the latter at 256 VGPRs) were slower than or level with 512 threads at depth 2.
    if (wj == 64) launch_sweep_t<64, 512>(s, uin, uout, rhs, ec, load_op, n, c, omega, r, r2);
    else launch_sweep_t<32, 256>(s, uin, uout, rhs, ec, load_op, n, c, omega, r, r2);
}

void sweep_rb_fused(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c,
                    double omega, IRange r, IRange r2)
{
    launch_sweep(s, uin, uout, rhs, uin, kLoadPlain, n, c, omega, r, r2);
}

void sweep_rb_fused_onload(hipStream_t s, View uin, View uout, View rhs, View ec, bool correct,
                           const int64_t n[3], Coef c, double omega, IRange r, IRange r2, bool outside_zero)
{
    const int op = correct ? kLoadCorrect : (outside_zero ? kLoadProlongZero : kLoadProlong);
    launch_sweep(s, uin, uout, rhs, ec, op, n, c, omega, r, r2);
}

}  // namespace mgk
