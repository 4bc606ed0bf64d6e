// PotentialSolver::computeEF -- host-side field gradient (reference cpp/PotentialSolver.cpp:418-458).
#include "PotentialSolver.h"

void PotentialSolver::computeEF()
{
    Field &phi = world.phi;
    double3 dh = world.getDh();
    const int n[3] = {world.ni, world.nj, world.nk};
    for (int i = 0; i < n[0]; i++)
        for (int j = 0; j < n[1]; j++)
            for (int k = 0; k < n[2]; k++) {
                const int at[3] = {i, j, k};
                double3 &e = world.ef[i][j][k];
                for (int a = 0; a < 3; a++) {
                    // value of phi displaced by s nodes along axis a
                    auto f = [&](int s) {
                        int q[3] = {i, j, k};
                        q[a] += s;
                        return phi[q[0]][q[1]][q[2]];
                    };
                    if (at[a] == 0) e[a] = -(-3 * f(0) + 4 * f(1) - f(2)) / (2 * dh[a]);
                    else if (at[a] == n[a] - 1) e[a] = -(f(-2) - 4 * f(-1) + 3 * f(0)) / (2 * dh[a]);
                    else e[a] = -(f(1) - f(-1)) / (2 * dh[a]);
                }
            }
}
