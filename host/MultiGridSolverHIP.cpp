// Host adapter: Field_<double> <-> flat arrays <-> libmg_hip (include/mg_hip.h).
#include "MultiGridSolverHIP.h"

#include <cstdint>
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

// True when the rows f[i][j] lie back to back in memory ((i*nj)+j)*nk+k -- the case for this
// repository's default Field, never for the reference's row-per-allocation Field
// (cpp/Field.h:65-70).
template <typename F>
bool rows_contiguous(F &f)
{
    const auto *base = f[0][0];
    for (int i = 0; i < f.ni; i++)
        for (int j = 0; j < f.nj; j++)
            if (f[i][j] != base + (std::size_t(i) * f.nj + j) * f.nk) return false;
    return true;
}

// Node p of the interior sample of the "world.phi is what the device holds" test: 4096 nodes spread
// by a multiplicative hash.
constexpr int kScattered = 4096;
void sample_node(const Field &f, int p, int &i, int &j, int &k)
{
    const std::uint64_t total = std::uint64_t(f.ni) * f.nj * f.nk;
    const std::uint64_t q = (std::uint64_t(p) * 0x9E3779B97F4A7C15ull >> 11) % total;
    k = int(q % f.nk);
    j = int(q / f.nk % f.nj);
    i = int(q / f.nk / f.nj);
}

// Calls fn(value) for every node of the six faces of f, in a fixed order (edge and corner nodes
// more than once).  The Dirichlet values of the solve live on these nodes (cpp/World.cpp:40-52).
template <typename F, typename Fn>
void for_each_face_node(F &f, Fn fn)
{
    for (int i = 0; i < f.ni; i += (f.ni > 1 ? f.ni - 1 : 1))
        for (int j = 0; j < f.nj; j++)
            for (int k = 0; k < f.nk; k++) fn(f[i][j][k]);
    for (int i = 0; i < f.ni; i++) {
        for (int j = 0; j < f.nj; j += (f.nj > 1 ? f.nj - 1 : 1))
            for (int k = 0; k < f.nk; k++) fn(f[i][j][k]);
        for (int j = 0; j < f.nj; j++) {
            fn(f[i][j][0]);
            fn(f[i][j][f.nk - 1]);
        }
    }
}

}  // namespace

// Dense [i][j][k] image of a field: the field's own storage if contiguous, else a row-wise copy
// into the page-locked staging block.
const double *MultiGridSolverHIP::pack(Field &f)
{
    if (rows_contiguous(f)) return f[0][0];
    for (int i = 0; i < f.ni; i++)
        for (int j = 0; j < f.nj; j++)
            std::memcpy(stage + (std::size_t(i) * f.nj + j) * f.nk, f[i][j], sizeof(double) * f.nk);
    return stage;
}

MultiGridSolverHIP::MultiGridSolverHIP(World &w, int max_it, double tol, int depth, int device, bool variant_b)
    : PotentialSolver(w), max_solver_it(max_it), tolerance(tol)
{
    init(depth, device, variant_b, {}, 0);
}

MultiGridSolverHIP::MultiGridSolverHIP(World &w, int max_it, double tol, int depth, const std::vector<int> &devices,
                                       int coarse_cut)
    : PotentialSolver(w), max_solver_it(max_it), tolerance(tol)
{
    if (devices.empty() || devices.size() > MG_MAX_GPUS) throw std::runtime_error("MultiGridSolverHIP: 1..16 devices");
    init(depth, devices[0], false, devices, coarse_cut);
}

void MultiGridSolverHIP::init(int depth, int device, bool variant_b, const std::vector<int> &devices, int coarse_cut)
{
    mg_config cfg;
    double3 dh = world.getDh();
    const double h[3] = {dh[0], dh[1], dh[2]};
    if (variant_b) check(mg_default_config_b(&cfg, world.ni, world.nj, world.nk, h, depth), "mg_default_config_b");
    else check(mg_default_config(&cfg, world.ni, world.nj, world.nk, h, depth), "mg_default_config");
    cfg.device = device;
    // computeEF() follows every solve() in the reference's loop: its array is part of the handle from
    // the start, not an allocation inside the first call
    if (devices.size() <= 1) cfg.flags |= MG_FLAG_RESERVE_EF;
    if (devices.size() > 1) {
        cfg.n_gpus = int(devices.size());
        for (std::size_t g = 0; g < devices.size(); g++) cfg.devices[g] = devices[g];
        cfg.coarse_cut = coarse_cut;
    }
    check(mg_create(&cfg, &handle), "mg_create");
    // Page-lock the fields in place where they are one block (failure only costs copy speed, e.g.
    // when another solver on the same World has locked them already); otherwise stage through a
    // page-locked block of our own, big enough for the three-component field.
    const std::size_t nodes = std::size_t(world.ni) * world.nj * world.nk;
    auto lock = [&](void *p, std::size_t bytes) {
        if (mg_host_register(p, bytes) == MG_OK) registered.push_back(p);
    };
    const bool flat_fields = rows_contiguous(world.phi) && rows_contiguous(world.rho) && rows_contiguous(world.ef);
    if (flat_fields) {
        lock(world.phi[0][0], nodes * sizeof(double));
        lock(world.rho[0][0], nodes * sizeof(double));
        lock(world.ef[0][0], nodes * 3 * sizeof(double));
    } else {
        stage_doubles = nodes * 3;
        void *p = nullptr;
        if (mg_host_alloc(&p, stage_doubles * sizeof(double)) != MG_OK) {
            mg_destroy(handle);
            throw std::runtime_error(std::string("mg_host_alloc: ") + mg_last_error());
        }
        stage = static_cast<double *>(p);
    }
}

MultiGridSolverHIP::~MultiGridSolverHIP()
{
    mg_destroy(handle);
    for (void *p : registered) mg_host_unregister(p);
    if (stage) mg_host_free(stage);
}

// Only with setAssumePhiUntouched(true): every face node (the Dirichlet values: a stale one would
// change the converged answer) and a scattered sample of the interior (a stale initial guess only
// costs iterations) must still hold what the last solve() returned.
bool MultiGridSolverHIP::device_phi_current()
{
    if (!assume_untouched || !device_phi_valid || phi_sample.size() != std::size_t(kScattered)) return false;
    Field &phi = world.phi;
    for (int p = 0; p < kScattered; p++) {
        int i, j, k;
        sample_node(phi, p, i, j, k);
        if (std::memcmp(&phi[i][j][k], &phi_sample[p], sizeof(double)) != 0) return false;
    }
    std::size_t at = 0;
    bool same = true;
    const std::vector<double> &kept = phi_faces;
    for_each_face_node(phi, [&](const double &v) {
        if (at >= kept.size() || std::memcmp(&v, &kept[at], sizeof(double)) != 0) same = false;
        at++;
    });
    return same && at == kept.size();
}

void MultiGridSolverHIP::remember_phi()
{
    device_phi_valid = false;
    if (!assume_untouched) return;  // the next call uploads whatever world.phi holds then
    Field &phi = world.phi;
    phi_sample.resize(kScattered);
    for (int p = 0; p < kScattered; p++) {
        int i, j, k;
        sample_node(phi, p, i, j, k);
        phi_sample[p] = phi[i][j][k];
    }
    phi_faces.clear();
    for_each_face_node(phi, [&](const double &v) { phi_faces.push_back(v); });
    device_phi_valid = true;
}

void MultiGridSolverHIP::upload_phi_if_stale()
{
    if (device_phi_current()) return;
    check(mg_set_phi(handle, pack(world.phi)), "mg_set_phi");
    phi_uploads++;
    remember_phi();
}

bool MultiGridSolverHIP::solve()
{
    int converged = 0;
    try {
        check(mg_set_rho(handle, pack(world.rho)), "mg_set_rho");
        upload_phi_if_stale();
        device_phi_valid = false;  // until the result is back in world.phi
        check(iterate(&converged), "solve");
        Field &phi = world.phi;
        if (rows_contiguous(phi)) {
            check(mg_get_phi(handle, phi[0][0]), "mg_get_phi");
        } else {
            check(mg_get_phi(handle, stage), "mg_get_phi");
            for (int i = 0; i < phi.ni; i++)
                for (int j = 0; j < phi.nj; j++)
                    std::memcpy(phi[i][j], stage + (std::size_t(i) * phi.nj + j) * phi.nk, sizeof(double) * phi.nk);
        }
        remember_phi();
    } catch (const std::runtime_error &e) {
        if (on_error == Throw) throw;
        std::cerr << "MultiGridSolverHIP::solve: " << e.what() << std::endl;
        return false;
    }
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

// E = -grad(phi) of the CURRENT world.phi, like the reference's host loop
// (cpp/PotentialSolver.cpp:418-458): computed on the device from the copy the last solve() left
// there when world.phi still is that potential, after an upload otherwise.
void MultiGridSolverHIP::computeEF()
{
    try {
        upload_phi_if_stale();
        check(mg_compute_ef(handle), "mg_compute_ef");
        download_ef();
    } catch (const std::runtime_error &e) {
        if (on_error == Throw) throw;
        std::cerr << "MultiGridSolverHIP::computeEF: " << e.what() << std::endl;
    }
}

void MultiGridSolverHIP::download_ef()
{
    Field3 &ef = world.ef;
    static_assert(sizeof(double3) == 3 * sizeof(double), "double3 must be three packed doubles");
    if (rows_contiguous(ef)) {
        check(mg_get_ef(handle, reinterpret_cast<double *>(ef[0][0])), "mg_get_ef");
        return;
    }
    const std::size_t row = std::size_t(ef.nk) * 3;
    check(mg_get_ef(handle, stage), "mg_get_ef");
    for (int i = 0; i < ef.ni; i++)
        for (int j = 0; j < ef.nj; j++)
            std::memcpy(static_cast<void *>(ef[i][j]), stage + (std::size_t(i) * ef.nj + j) * row, sizeof(double) * row);
}

double MultiGridSolverHIP::deviceMilliseconds() const
{
    double ms = 0;
    mg_last_call_ms(handle, &ms);
    return ms;
}
