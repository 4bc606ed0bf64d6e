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

#include <cstdlib>

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

// WJ rows x 2*WKP columns per window; outputs are the window minus two rings.
// D = planes of u (and of the right-hand side) requested ahead of use: the march is
// latency bound unless several planes per workgroup are in flight.
template <int WJ, int WKP, int D, int NT>
__global__ __launch_bounds__(NT) void k_sweep_rb(View uin, View uout, View rhs, int nI, int nJ,
                                                  int nK, Coef c, double omega, int CI, int nchunks,
                                                  int tiles_k, int rlo, int rhi, int xcd_map)
{
    constexpr int WK = 2 * WKP;
    constexpr int TJ = WJ - 4, TK = WK - 4;
    constexpr int RPP = NT / WKP;   // rows covered by one pass of the workgroup
    constexpr int Q = WJ / RPP;     // pair-columns owned by a thread
    static_assert(WJ % RPP == 0, "window rows must be a multiple of rows per pass");
    __shared__ double lds[3][WJ][WK];

    const int tid = threadIdx.x;
    const int m = tid % WKP, rr = tid / WKP;
    // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 labels the XCD).  Remap so that
    // each XCD works on a contiguous run of (chunk, tile) items: windows that share halo columns and
    // rows then run on one XCD at about the same time, and the second fetch of a halo line hits
    // that XCD's L2.  Speed only: any mapping gives the same result.
    int chunk, tile;
    if (xcd_map) {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7;
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int w = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + idx;
        const int ntiles = nwg / nchunks;
        chunk = w / ntiles;
        tile = w % ntiles;
    } else {
        chunk = blockIdx.x % nchunks;
        tile = blockIdx.x / nchunks;
    }
    const int tk = tile % tiles_k, tj = tile / tiles_k;
    const int j0 = tj * TJ - 2, k0 = tk * TK - 2;
    const int uJ = (int)uin.d[1], uK = (int)uin.d[2];
    const int rJ = (int)rhs.d[1], rK = (int)rhs.d[2];
    // stored plane ranges of the (possibly slab) views, global indices
    const int ulo = (int)uin.i0, uhi = (int)uin.i_end() - 1;
    const int blo = (int)rhs.i0, bhi = (int)rhs.i_end() - 1;
    const int ia = rlo + chunk * CI;
    const int ib = min(ia + CI, rhi);
    if (ia >= ib) return;  // uniform per workgroup
    const int kk = k0 + 2 * m;  // first column of this lane's pairs

    // Loads are unconditional with clamped coordinates: a branch around a load makes
    // the compiler wait for it on the spot, which serialises the whole prefetch.
    // Clamped (out-of-array) lanes fetch some valid element that is never used.
    const int kku = min(max(kk, 0), (uK - 1) & ~1);
    const int kkr = min(max(kk, 0), (rK - 1) & ~1);
    auto load_u = [&](int i, int q) -> d2 {
        const int j = min(max(j0 + q * RPP + rr, 0), uJ - 1);
        const int ic = min(max(i, ulo), uhi);
        return *reinterpret_cast<const d2 *>(uin.p + uin.at(ic, j, kku));
    };
    auto load_b = [&](int i, int q) -> d2 {
        const int j = min(max(j0 + q * RPP + rr, 0), rJ - 1);
        const int ic = min(max(i, blo), bhi);
        return *reinterpret_cast<const d2 *>(rhs.p + rhs.at(ic, j, kkr));
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
        const int r = q * RPP + rr;
        *reinterpret_cast<d2 *>(&lds[(ia - 1 + 3) % 3][r][2 * m]) = v0[q];
    }

    for (int p = ia - 1; p <= ib; p++) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            vf[D - 1][q] = load_u(min(p + 1 + D, ib + 1), q);
            bf[D - 1][q] = load_b(min(p + D, ib), q);
        }
        __syncthreads();
        // ---- red points of plane p (neighbours: old black values) ----------
        if (p >= 1 && p <= nI - 2) {
            double(*X)[WK] = lds[(p + 3) % 3];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int r = q * RPP + rr;
                const int j = j0 + r;
                const int e = (p + j) & 1;
                const int col = 2 * m + e;
                const int k = k0 + col;
                const bool ok = j >= 1 && j <= nJ - 2 && k >= 1 && k <= nK - 2 && r >= 1 &&
                                r <= WJ - 2 && col >= 1 && col <= WK - 2;
                const int rn = max(r - 1, 0), rs = min(r + 1, WJ - 1);
                const int cw = max(col - 1, 0), ce = min(col + 1, WK - 1);
                const double uc = pick(v0[q], e);
                const double g = div_d(pick(b0[q], e) + c.cx * (pick(vm1[q], e) + pick(vp1[q], e)) +
                                       c.cy * (X[rn][col] + X[rs][col]) + c.cz * (X[r][cw] + X[r][ce]), c);
                const double un = ok ? uc + omega * (g - uc) : uc;
                v0[q] = put(v0[q], e, un);
                X[r][col] = un;
            }
        }
        __syncthreads();
        // ---- black points of plane p-1 (neighbours: new red values), store --
        const int pb = p - 1;
        if (pb >= ia && pb < ib) {
            if (pb >= 1 && pb <= nI - 2) {
                double(*Y)[WK] = lds[(pb + 3) % 3];
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int r = q * RPP + rr;
                    const int j = j0 + r;
                    const int e = (p + j) & 1;  // black of plane p-1 sits where red of plane p does
                    const int col = 2 * m + e;
                    const int k = k0 + col;
                    const bool ok = j >= 1 && j <= nJ - 2 && k >= 1 && k <= nK - 2 && r >= 2 &&
                                    r <= WJ - 3 && col >= 2 && col <= WK - 3;
                    const int rn = max(r - 1, 0), rs = min(r + 1, WJ - 1);
                    const int cw = max(col - 1, 0), ce = min(col + 1, WK - 1);
                    const double uc = pick(vm1[q], e);
                    const double g = div_d(pick(bm1[q], e) + c.cx * (pick(vm2[q], e) + pick(v0[q], e)) +
                                           c.cy * (Y[rn][col] + Y[rs][col]) + c.cz * (Y[r][cw] + Y[r][ce]), c);
                    vm1[q] = put(vm1[q], e, ok ? uc + omega * (g - uc) : uc);
                }
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int r = q * RPP + rr;
                const int j = j0 + r;
                if (r >= 2 && r <= WJ - 3 && m >= 1 && m <= WKP - 2 && j < uJ && kk < uK)
                    *reinterpret_cast<d2 *>(uout.p + uout.at(pb, j, kk)) = vm1[q];
            }
        }
        // plane p+1 (old values) into the LDS slot plane p-2 occupied
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int r = q * RPP + rr;
            *reinterpret_cast<d2 *>(&lds[(p + 1 + 3) % 3][r][2 * m]) = vp1[q];
            vm2[q] = vm1[q];
            vm1[q] = v0[q];
            v0[q] = vp1[q];
            vp1[q] = vf[0][q];
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

}  // namespace

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// One full red-black sweep u_in -> u_out: planes r of u_out, updated over 1..n-2 per axis
// (n = level size); every other element of those planes is copied through unchanged.
void sweep_rb_fused(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c,
                    double omega, IRange r)
{
    constexpr int WJ = 32, WKP = 32;
    constexpr int TJ = WJ - 4, TK = 2 * WKP - 4;
    static const int depth = env_int("MG_SWEEP_D", 2);         // tuning knobs (A/B runs)
    static const int want_chunks = env_int("MG_SWEEP_CHUNKS", 0);
    const int64_t planes = r.hi - r.lo;
    if (planes <= 0) return;
    const int tiles_j = (int)((uin.d[1] + TJ - 1) / TJ);
    const int tiles_k = (int)((uin.d[2] + TK - 1) / TK);
    static const int xcd_map = env_int("MG_SWEEP_MAP", 0);
    const int tiles = tiles_j * tiles_k;
    int nchunks;
    if (xcd_map) {
        // one cohort: about two workgroups per CU resident at once, all started together
        nchunks = (int)((512 + tiles / 2) / tiles);
        if (nchunks < 1) nchunks = 1;
        while (nchunks > 1 && planes / nchunks < 8) nchunks--;
    } else {
        nchunks = pick_chunks(tiles, planes, 4);
    }
    if (want_chunks > 0) nchunks = want_chunks;
    const int CI = (int)((planes + nchunks - 1) / nchunks);
    const unsigned grid = (unsigned)(nchunks * tiles);
    static const int threads = env_int("MG_SWEEP_NT", 256);
#define LAUNCH(D, NT)                                                                              \
    hipLaunchKernelGGL((k_sweep_rb<WJ, WKP, D, NT>), dim3(grid), dim3(NT), 0, s, uin, uout, rhs,   \
                       (int)n[0], (int)n[1], (int)n[2], c, omega, CI, nchunks, tiles_k, (int)r.lo, \
                       (int)r.hi, xcd_map)
    if (threads == 512) {
        if (depth <= 1) LAUNCH(1, 512);
        else if (depth == 2) LAUNCH(2, 512);
        else LAUNCH(3, 512);
    } else {
        if (depth <= 1) LAUNCH(1, 256);
        else if (depth == 2) LAUNCH(2, 256);
        else LAUNCH(3, 256);
    }
#undef LAUNCH
}

}  // namespace mgk
