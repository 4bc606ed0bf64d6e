// Red-black GS-SOR sweep by 3-D tiles: both colours in ONE launch, out of place, for the arrays
// the marching kernels of mg_sweep.hip serve badly -- levels with fewer than 128 rows (where a
// colour-per-launch smoother costs two dependent launches per sweep and each launch is pure
// latency) and plane ranges too short to march along.
//
// A workgroup owns a TI x TJ x TK block of points.  It loads the block plus two rings of u_in,
// updates the red points of the block plus ONE ring (their black neighbours are old values, all
// inside the loaded region), then, after a barrier, the black points of the block itself (their red
// neighbours are the values just computed), and stores the block to u_out.  The red values of the
// ring are computed redundantly by the neighbouring workgroups -- from the same operands with the
// same expression, hence to the same bits -- which is what removes the kernel boundary between the
// colours.  Points outside 1..n-2 are copied through.
//
// Thread mapping: the loaded region is BI x 16 x 32 (BI = TI + 4, 16 = TJ + 4 rows, 32 = TK + 4
// columns = 16 aligned pairs).  Thread (row, pair) of the 256 keeps its pair of every one of the BI
// planes in registers: the i-1 / i+1 neighbours of a point are the thread's own registers, the
// other element of the pair is its k-1 or k+1 neighbour, and only the j-1 / j+1 neighbours and the
// adjacent pair's element come from the LDS image, whose two pair elements are stored apart
// ([element][plane][row][pair]: consecutive lanes read consecutive words).  All index arithmetic is
// shifts and masks.
//
// Two sweeps per launch (S = 2): the same with four rings -- red of sweep 1 on the block plus three
// rings, black of sweep 1 on the block plus two, red of sweep 2 on the block plus one, black of sweep 2
// on the block, a barrier between the phases.  A launch then owns less of what it loads (8 x 24 of
// 16 x 32 per plane, TI of TI + 8 planes), which pays where a launch is latency and not work: the
// levels of 65^3 and below, whose five sweeps become three launches.
//
// The update is the reference's expression evaluated in its order
// (cpp/PotentialSolver.cpp:3780-3786, :3929-3935): results are bit-identical to the
// colour-per-launch kernel and to the marching kernels.
#include "mg_kernels.h"

#include <type_traits>

namespace mgk {

namespace {

struct alignas(16) d2 { double x, y; };

__device__ __forceinline__ double pick(d2 v, int e) { return e ? v.y : v.x; }

// owned rows / columns of a tile that does S sweeps: the loaded 16 rows x 16 pairs minus 2 S rings
constexpr int tile_j(int S) { return 16 - 4 * S; }
constexpr int tile_k(int S) { return 32 - 4 * S; }

template <int TI, bool FAST, int S = 1>
__global__ __launch_bounds__(256) void k_sweep_tile(View uin, View uout, View rhs, int nI, int nJ, int nK, Coef c,
                                                     double omega, int tiles_j, int tiles_k, int tiles_i1, int lo1,
                                                     int hi1, int lo2, int hi2)
{
    constexpr int RINGS = 2 * S, kTileJ = tile_j(S), kTileK = tile_k(S);
    constexpr int BI = TI + 2 * RINGS, BJ = kTileJ + 2 * RINGS, H = (kTileK + 2 * RINGS) / 2;
    static_assert(BJ == 16 && H == 16, "thread mapping: 16 rows of 16 pairs");
    static_assert(S == 1 || S == 2, "one or two sweeps per launch");
    constexpr int PLANE = BJ * H;            // pairs per plane of the region = threads
    constexpr int ELEM = BI * PLANE + 2;     // one pair element of the region, a guard word at either end
    __shared__ double su[2 * ELEM];

    const int tid = threadIdx.x;
    const int lkp = tid & 15, lj = tid >> 4;
    int b = blockIdx.x;
    const int tk = b % tiles_k;
    b /= tiles_k;
    const int tj = b % tiles_j;
    int ti = b / tiles_j;
    int rlo = lo1, rhi = hi1;
    if (ti >= tiles_i1) {  // second plane range of the launch
        ti -= tiles_i1;
        rlo = lo2;
        rhi = hi2;
    }
    const int gi0 = rlo + ti * TI - RINGS, gj = tj * kTileJ - RINGS + lj, gk = tk * kTileK - RINGS + 2 * lkp;  // gk: the pair's even element
    const int own_hi = min(rlo + ti * TI + TI, rhi);  // owned planes: [gi0 + RINGS, own_hi)
    const int uJ = (int)uin.d[1], uK = (int)uin.d[2];
    const int rJ = (int)rhs.d[1], rK = (int)rhs.d[2];
    const int ulo = (int)uin.i0, uhi = (int)uin.i_end() - 1;
    const int blo = (int)rhs.i0, bhi = (int)rhs.i_end() - 1;

    // Every global load is requested before anything waits, unconditionally and with clamped
    // coordinates (a branch around a load makes the compiler wait for it on the spot); a point that
    // is not updated never uses what was fetched for it.
    d2 v[BI], rb[BI];
    {
        const int64_t ou = (int64_t)min(max(gj, 0), uJ - 1) * uin.sj + min(max(gk, 0), (uK - 1) & ~1);
        const int64_t ob = (int64_t)min(max(gj, 0), rJ - 1) * rhs.sj + min(max(gk, 0), (rK - 1) & ~1);
#pragma unroll
        for (int q = 0; q < BI; q++)
            v[q] = *reinterpret_cast<const d2 *>(uin.p + (int64_t)(min(max(gi0 + q, ulo), uhi) - ulo) * uin.si + ou);
#pragma unroll
        for (int q = 1; q < BI - 1; q++)
            rb[q] = *reinterpret_cast<const d2 *>(rhs.p + (int64_t)(min(max(gi0 + q, blo), bhi) - blo) * rhs.si + ob);
    }
    // which points of this thread's column of pairs may be updated at all (interior of the level):
    // element x at gk, y at gk + 1
    const bool jin = gj >= 1 && gj <= nJ - 2;
    const bool xin = jin && gk >= 1 && gk <= nK - 2, yin = jin && gk + 1 >= 1 && gk + 1 <= nK - 2;
    // phase p (0: red of the first sweep, 1: its black, ...) updates the region minus p + 1 rings: rows
    // and columns p+1 .. 14-p / 30-p of the region (the pair's x at column 2 lkp, y at 2 lkp + 1)
    const auto xok = [&](int ring) { return xin && lj >= ring && lj <= BJ - 1 - ring && 2 * lkp >= ring && 2 * lkp <= 2 * H - 1 - ring; };
    const auto yok = [&](int ring) { return yin && lj >= ring && lj <= BJ - 1 - ring && 2 * lkp + 1 >= ring && 2 * lkp + 1 <= 2 * H - 1 - ring; };
    // the block itself: the region minus all rings (whole pairs: RINGS is even)
    const bool own = lj >= RINGS && lj <= BJ - 1 - RINGS && lkp >= RINGS / 2 && lkp <= H - 1 - RINGS / 2;

    double *X = su + 1, *Y = su + ELEM + 1;   // element images; the guard words take the reads off either end
    const int at = lj * H + lkp;
#pragma unroll
    for (int q = 0; q < BI; q++) {
        X[q * PLANE + at] = v[q].x;
        Y[q * PLANE + at] = v[q].y;
    }
    __syncthreads();
    // offsets of the row above / below, clamped for the threads of the first and last row (they
    // update nothing; the clamp keeps their reads inside the image)
    const int up = lj > 0 ? -H : 0, dn = lj < BJ - 1 ? H : 0;

    // One colour of the planes [QA, QB): the element of plane q's pair whose global colour is
    // `colour`, where xok / yok allows it.  publish: the new value also goes to the image.
    auto phase = [&](auto qa_, auto qb_, int colour, bool xok, bool yok, bool publish) {
        constexpr int QA = decltype(qa_)::value, QB = decltype(qb_)::value, N = QB - QA;
        double bb[N], si[N], sj[N], sk[N], u0[N], g[N];
        int e[N];
#pragma unroll
        for (int t = 0; t < N; t++) {
            const int q = QA + t;
            e[t] = (gi0 + q + gj + colour) & 1;                  // 1: the odd element has this colour
            const double *E = (e[t] ? Y : X) + q * PLANE + at;   // this element's image: j neighbours
            const double *F = (e[t] ? X : Y) + q * PLANE + at;   // the other element's: the far k neighbour
            const double mate = pick(v[q], 1 - e[t]);
            const double far = e[t] ? F[1] : F[-1];
            bb[t] = pick(rb[q], e[t]);
            si[t] = pick(v[q - 1], e[t]) + pick(v[q + 1], e[t]);
            sj[t] = E[up] + E[dn];
            sk[t] = e[t] ? mate + far : far + mate;              // u[k-1] + u[k+1]
            u0[t] = pick(v[q], e[t]);
        }
        sor_n<FAST, N>(bb, si, sj, sk, u0, c, omega, g);
#pragma unroll
        for (int t = 0; t < N; t++) {
            const int q = QA + t;
            const int gi = gi0 + q;
            const bool ok = (e[t] ? yok : xok) && gi >= 1 && gi <= nI - 2;
            const double un = ok ? g[t] : u0[t];
            if (e[t]) v[q].y = un;
            else v[q].x = un;
            if (publish) (e[t] ? Y : X)[q * PLANE + at] = un;
        }
    };
    phase(std::integral_constant<int, 1>{}, std::integral_constant<int, BI - 1>{}, 0, xok(1), yok(1), true);
    __syncthreads();
    phase(std::integral_constant<int, 2>{}, std::integral_constant<int, BI - 2>{}, 1, xok(2), yok(2), S > 1);
    if constexpr (S > 1) {
        __syncthreads();
        phase(std::integral_constant<int, 3>{}, std::integral_constant<int, BI - 3>{}, 0, xok(3), yok(3), true);
        __syncthreads();
        phase(std::integral_constant<int, 4>{}, std::integral_constant<int, BI - 4>{}, 1, xok(4), yok(4), false);
    }

    if (own && gj < uJ && gk < uK) {
        double *o = uout.p + (int64_t)gj * uout.sj + gk;
#pragma unroll
        for (int q = RINGS; q < BI - RINGS; q++) {
            const int gi = gi0 + q;
            if (gi < own_hi) *reinterpret_cast<d2 *>(o + (int64_t)(gi - (int)uout.i0) * uout.si) = v[q];
        }
    }
}

template <int TI, int S = 1>
void launch_tile(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega, IRange r,
                 IRange r2)
{
    constexpr int kTileJ = tile_j(S), kTileK = tile_k(S);
    const int tiles_j = (int)((uin.d[1] + kTileJ - 1) / kTileJ), tiles_k = (int)((uin.d[2] + kTileK - 1) / kTileK);
    const int tiles_i1 = (int)((r.hi - r.lo + TI - 1) / TI);
    const int tiles_i2 = r2.hi > r2.lo ? (int)((r2.hi - r2.lo + TI - 1) / TI) : 0;
    const unsigned grid = (unsigned)(tiles_j * tiles_k) * (unsigned)(tiles_i1 + tiles_i2);
    if (c.fast)
        hipLaunchKernelGGL((k_sweep_tile<TI, true, S>), dim3(grid), dim3(256), 0, s, uin, uout, rhs, (int)n[0], (int)n[1],
                           (int)n[2], c, omega, tiles_j, tiles_k, tiles_i1, (int)r.lo, (int)r.hi, (int)r2.lo, (int)r2.hi);
    else
        hipLaunchKernelGGL((k_sweep_tile<TI, false, S>), dim3(grid), dim3(256), 0, s, uin, uout, rhs, (int)n[0], (int)n[1],
                           (int)n[2], c, omega, tiles_j, tiles_k, tiles_i1, (int)r.lo, (int)r.hi, (int)r2.lo, (int)r2.hi);
}

}  // namespace

// One full red-black sweep of planes r (and r2) of u_in into u_out by 3-D tiles.  u_in must hold
// planes [r.lo-2, r.hi+2) as far as they exist in the level, rhs planes [r.lo-1, r.hi+1).
void sweep_rb_tile(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega, IRange r,
                   IRange r2)
{
    if (r.hi <= r.lo) return;
    static const int shape = exp_knob("MG_EXP_TILE", 0);
    const int64_t planes = r.hi - r.lo;
    // planes per tile: few for short ranges and small levels (more workgroups), more for large
    // planes (each tile re-reads four planes of halo)
    // (measured per sweep with 2 / 4 / 8 planes per tile, 34^3: 3.8 / 4.4 / 5.5 us; 66^3: 7.3 - 7.5 us
    // alike; 40 planes of 129^2: 10.0 / 9.1 / 9.7)
    const int64_t area = uin.d[1] * uin.d[2];
    int ti = (planes <= 2 || area <= 40 * 40) ? 2 : (planes <= 4 || area <= 136 * 136) ? 4 : 8;
    if (shape == 2 || shape == 4 || shape == 8) ti = shape;
    if (ti == 2) launch_tile<2>(s, uin, uout, rhs, n, c, omega, r, r2);
    else if (ti == 4) launch_tile<4>(s, uin, uout, rhs, n, c, omega, r, r2);
    else launch_tile<8>(s, uin, uout, rhs, n, c, omega, r, r2);
}

// TWO full red-black sweeps of planes r of u_in into u_out by 3-D tiles in one launch.  u_in must hold
// planes [r.lo-4, r.hi+4) as far as they exist in the level, rhs planes [r.lo-3, r.hi+3).
void sweep_rb_tile2(hipStream_t s, View uin, View uout, View rhs, const int64_t n[3], Coef c, double omega, IRange r)
{
    if (r.hi <= r.lo) return;
    static const int shape = exp_knob("MG_EXP_TILE2", 0);
    // planes per tile, by measurement as for the single sweep
    int ti = (r.hi - r.lo <= 2 || uin.d[1] * uin.d[2] <= 40 * 40) ? 2 : 4;
    if (shape == 2 || shape == 4) ti = shape;
    if (ti == 2) launch_tile<2, 2>(s, uin, uout, rhs, n, c, omega, r, IRange{0, 0});
    else launch_tile<4, 2>(s, uin, uout, rhs, n, c, omega, r, IRange{0, 0});
}

}  // namespace mgk
