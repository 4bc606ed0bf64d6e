// Host adapter: Field_<double> <-> flat arrays <-> libmg_hip (include/mg_hip.h).
#include "MultiGridSolverHIP.h"

#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>

#include "../include/mg_hip.h"

namespace {

void check(int rc, const char *what)
{
    if (rc != MG_OK) throw std::runtime_error(std::string(what) + ": " + mg_last_error());
}

// True when the rows f[i][j] lie back to back in memory ((i*nj)+j)*nk+k -- always the case for
// this repository's Field, never for the reference's row-per-allocation Field.
bool rows_contiguous(Field &f)
{
    const double *base = f[0][0];
    for (int i = 0; i < f.ni; i++)
        for (int j = 0; j < f.nj; j++)
            if (f[i][j] != base + (std::size_t(i) * f.nj + j) * f.nk) return false;
    return true;
}

// Dense [i][j][k] image of a field: the field's own storage if contiguous, else a row-wise copy.
const double *pack(Field &f, std::vector<double> &stage)
{
    if (rows_contiguous(f)) return f[0][0];
    stage.resize(std::size_t(f.ni) * f.nj * f.nk);
    for (int i = 0; i < f.ni; i++)
        for (int j = 0; j < f.nj; j++)
            std::memcpy(&stage[(std::size_t(i) * f.nj + j) * f.nk], f[i][j], sizeof(double) * f.nk);
    return stage.data();
}

}  // namespace

MultiGridSolverHIP::MultiGridSolverHIP(World &w, int max_it, double tol, int depth, int device, bool variant_b)
    : PotentialSolver(w), max_solver_it(max_it), tolerance(tol)
{
    mg_config cfg;
    double3 dh = world.getDh();
    const double h[3] = {dh[0], dh[1], dh[2]};
    if (variant_b) check(mg_default_config_b(&cfg, world.ni, world.nj, world.nk, h, depth), "mg_default_config_b");
    else check(mg_default_config(&cfg, world.ni, world.nj, world.nk, h, depth), "mg_default_config");
    cfg.device = device;
    check(mg_create(&cfg, &handle), "mg_create");
}

MultiGridSolverHIP::~MultiGridSolverHIP() { mg_destroy(handle); }

bool MultiGridSolverHIP::solve()
{
    check(mg_set_rho(handle, pack(world.rho, stage)), "mg_set_rho");
    check(mg_set_phi(handle, pack(world.phi, stage)), "mg_set_phi");
    int converged = 0;
    check(iterate(&converged), "solve");
    Field &phi = world.phi;
    if (rows_contiguous(phi)) {
        check(mg_get_phi(handle, phi[0][0]), "mg_get_phi");
    } else {
        stage.resize(std::size_t(phi.ni) * phi.nj * phi.nk);
        check(mg_get_phi(handle, stage.data()), "mg_get_phi");
        for (int i = 0; i < phi.ni; i++)
            for (int j = 0; j < phi.nj; j++)
                std::memcpy(phi[i][j], &stage[(std::size_t(i) * phi.nj + j) * phi.nk], sizeof(double) * phi.nk);
    }
    solved = true;
    if (!converged) std::cerr << "GS failed to converge, L2=" << last_l2 << std::endl;
    return converged != 0;
}

int MultiGridSolverHIP::iterate(int *converged)
{
    return mg_solve(handle, max_solver_it, tolerance, converged, &last_cycles, &last_l2);
}

int GaussSeidelSolverHIP::iterate(int *converged)
{
    return mg_gs_solve(handle, max_solver_it, tolerance, converged, &last_cycles, &last_l2);
}

// E = -grad(phi) from the device-resident potential.  Before the first solve() the device holds
// no potential yet: upload world.phi first.
void MultiGridSolverHIP::computeEF()
{
    if (!solved) check(mg_set_phi(handle, pack(world.phi, stage)), "mg_set_phi");
    check(mg_compute_ef(handle), "mg_compute_ef");
    Field3 &ef = world.ef;
    static_assert(sizeof(double3) == 3 * sizeof(double), "double3 must be three packed doubles");
    const std::size_t row = std::size_t(ef.nk) * 3;
    const double *base = reinterpret_cast<const double *>(ef[0][0]);
    bool contiguous = true;
    for (int i = 0; i < ef.ni && contiguous; i++)
        for (int j = 0; j < ef.nj; j++)
            if (reinterpret_cast<const double *>(ef[i][j]) != base + (std::size_t(i) * ef.nj + j) * row) {
                contiguous = false;
                break;
            }
    if (contiguous) {
        check(mg_get_ef(handle, reinterpret_cast<double *>(ef[0][0])), "mg_get_ef");
        return;
    }
    stage.resize(std::size_t(ef.ni) * ef.nj * row);
    check(mg_get_ef(handle, stage.data()), "mg_get_ef");
    for (int i = 0; i < ef.ni; i++)
        for (int j = 0; j < ef.nj; j++)
            std::memcpy(static_cast<void *>(ef[i][j]), &stage[(std::size_t(i) * ef.nj + j) * row], sizeof(double) * row);
}

double MultiGridSolverHIP::deviceMilliseconds() const
{
    double ms = 0;
    mg_last_call_ms(handle, &ms);
    return ms;
}
