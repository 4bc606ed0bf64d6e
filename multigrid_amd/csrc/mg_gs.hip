// Convergence test of the plain Gauss-Seidel/SOR solver (reference GaussSeidelSolver::solve,
// cpp/PotentialSolver.cpp:267-286): sum of R^2 over the INTERIOR nodes only, with
//   R = -phi*d + rho/EPS_0 + cx*(phi[i-1]+phi[i+1]) + cy*(phi[j-1]+phi[j+1]) + cz*(phi[k-1]+phi[k+1])
// evaluated left to right (opposite sign and different association from the multigrid residual).
// The sweeps themselves are the multigrid smoother kernels.
#include "mg_kernels.h"

namespace mgk {

namespace {

constexpr int GS_BX = 64, GS_BY = 4;

__global__ __launch_bounds__(256) void k_gs_residual_sumsq(View phi, View b, int64_t ni, Coef c,
                                                           double *__restrict__ partials, int64_t ilo)
{
    __shared__ double lds[4];
    const int64_t nj = phi.d[1], nk = phi.d[2];
    const int64_t k = (int64_t)blockIdx.x * GS_BX + threadIdx.x;
    const int64_t j = (int64_t)blockIdx.y * GS_BY + threadIdx.y;
    const int64_t i = ilo + blockIdx.z;
    double r = 0;
    if (i >= 1 && i <= ni - 2 && j >= 1 && j <= nj - 2 && k >= 1 && k <= nk - 2) {
        const double *p = phi.p + phi.at(i, j, k);
        r = -p[0] * c.d + b.p[b.at(i, j, k)] + c.cx * (p[-phi.si] + p[phi.si]) +
            c.cy * (p[-phi.sj] + p[phi.sj]) + c.cz * (p[-1] + p[1]);
    }
    const double t = block_sum(r * r, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        partials[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

dim3 gs_grid(const int64_t n[3], IRange r)
{
    return dim3((unsigned)((n[2] + GS_BX - 1) / GS_BX), (unsigned)((n[1] + GS_BY - 1) / GS_BY),
                (unsigned)(r.hi - r.lo));
}

}  // namespace

int64_t gs_residual_partials(const int64_t n[3], IRange r)
{
    if (r.hi <= r.lo) return 0;
    const dim3 g = gs_grid(n, r);
    return (int64_t)g.x * g.y * g.z;
}

void gs_residual_sumsq(hipStream_t s, View phi, View b, const int64_t n[3], Coef c, double *partials, IRange r)
{
    if (r.hi <= r.lo) return;
    hipLaunchKernelGGL(k_gs_residual_sumsq, gs_grid(n, r), dim3(GS_BX, GS_BY), 0, s, phi, b, n[0], c, partials,
                       r.lo);
}

}  // namespace mgk
