#include "World.h"

World::World(int ni_, int nj_, int nk_)
    : ni(ni_), nj(nj_), nk(nk_), nn(ni_, nj_, nk_), phi(ni_, nj_, nk_), rho(ni_, nj_, nk_),
      node_vol(ni_, nj_, nk_), ef(ni_, nj_, nk_), object_id(ni_, nj_, nk_)
{
}

void World::setExtents(const double3 lo, const double3 hi)
{
    x0 = lo;
    xm = hi;
    for (int a = 0; a < 3; a++) {
        dh[a] = (xm(a) - x0(a)) / (nn(a) - 1);
        xc[a] = 0.5 * (x0(a) + xm(a));
    }
    computeNodeVolumes();
}

void World::computeNodeVolumes()
{
    auto wall = [](int v, int n) { return v == 0 || v == n - 1; };
    for (int i = 0; i < ni; i++)
        for (int j = 0; j < nj; j++)
            for (int k = 0; k < nk; k++) {
                double v = dh[0] * dh[1] * dh[2];
                if (wall(i, ni)) v *= 0.5;
                if (wall(j, nj)) v *= 0.5;
                if (wall(k, nk)) v *= 0.5;
                node_vol[i][j][k] = v;
            }
}

double World::getPE()
{
    double pe = 0;
    for (int i = 0; i < ni; i++)
        for (int j = 0; j < nj; j++)
            for (int k = 0; k < nk; k++) {
                double3 e = ef[i][j][k];
                pe += (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) * node_vol[i][j][k];
            }
    return 0.5 * Const::EPS_0 * pe;
}

void World::setBoundaries()
{
    auto wall = [](int v, int n) { return v == 0 || v == n - 1; };
    for (int i = 0; i < ni; i++)
        for (int j = 0; j < nj; j++)
            for (int k = 0; k < nk; k++)
                if (wall(i, ni) || wall(j, nj) || wall(k, nk)) {
                    object_id[i][j][k] = 1;
                    phi[i][j][k] = 0;
                }
}
