// HIP-backed multigrid solvers behind the reference's PotentialSolver interface.
//
//   MultiGridSolverHIP             depth-generic V-cycle solver on an MI355X through libmg_hip
//   GaussSeidelSolverHIP           the plain SOR loop of GaussSeidelSolver (cpp/PotentialSolver.cpp:235-291)
//   hip::MultiGridSolver, V2..V5,  same constructor signature (World&, int max_it, double tol) as
//   hip::GaussSeidelSolver         the reference classes of those names
//                                  (cpp/PotentialSolver.h:70-87, :171-272), with their sweep schedules
//
// computeEF() of these classes differentiates the potential the solve left in device memory
// (one kernel, no upload) and copies E into world.ef; results are bit-identical to the host loop.
//
// solve() contract (unchanged from the reference): reads world.phi (initial guess and
// Dirichlet face values), world.rho, world.getDh(); overwrites world.phi; returns true iff
// the residual norm fell below `tol` within `max_it` iterations, otherwise prints
// "GS failed to converge, L2=<value>" to stderr and returns false
// (cpp/PotentialSolver.cpp:4246-4247).  Solver-private work fields persist between calls.
// There is no CPU fallback: if the device library reports an error, solve() throws
// std::runtime_error with the library's message (the reference's solve() never throws and its
// driver has no handler, so the program stops with that message); setErrorPolicy(ReturnFalse)
// makes solve() print the message to stderr and return false instead.
//
// Host <-> device traffic of a solve(): rho and phi up, phi down -- solve() and computeEF() read the
// CURRENT world.phi in full on every call, as the reference does (its Fields live in host memory):
// an electrode patch written onto a face, or any other edit between two solves, is always seen.
// A driver that never writes world.phi between solves (the reference's time loop does not,
// cpp/Main.cpp:151-166) can save the upload with setAssumePhiUntouched(true): the potential is then
// uploaded only on the first call, after invalidate(), or when world.phi no longer matches what the
// last solve() returned on ALL nodes of the six faces (where the Dirichlet values live,
// cpp/World.cpp:40-52) and on 4096 scattered interior nodes; an edit of a few interior nodes alone
// (which can only cost iterations, never change the converged answer) must be announced with
// invalidate() in that mode.  Fields whose rows are contiguous are page-locked in place for the
// solver's lifetime, so the copies run at the full host-link rate.
//
// This header and MultiGridSolverHIP.cpp use only the public surface of World and Field
// (phi, rho, ef, getDh(), ni/nj/nk, f[i][j] row pointers), so they also compile against the
// reference's own World.h / Field.h / PotentialSolver.h, whose computeEF() is not virtual
// (cpp/PotentialSolver.h:63): computeEF() below then hides the base version (calls through a
// PotentialSolver* reach the host loop, which works on the downloaded phi); where the base
// declares it virtual, as this repository's PotentialSolver.h does, it overrides it.
#ifndef MG_HOST_MULTIGRID_SOLVER_HIP_H
#define MG_HOST_MULTIGRID_SOLVER_HIP_H

#include <vector>

#include "PotentialSolver.h"

struct mg_handle;

class MultiGridSolverHIP : public PotentialSolver {
public:
    // depth = number of coarse grids (1..5 = the reference's MultiGridSolver, V2, .. V5;
    // 0 only through GaussSeidelSolverHIP)
    // variant_b: the reference's "B" classes (cpp/PotentialSolver.h:274-350): 7-point restriction,
    // and for depth 5 the sweep schedule 1,2,3,4,5,50
    MultiGridSolverHIP(World &world, int max_it, double tol, int depth, int device = 0, bool variant_b = false);
    // The grid split into slabs of i-planes over several GPUs (one host thread per GPU inside the
    // library, halo planes over RCCL; an ordinal may repeat, e.g. {0, 0}: two slabs on one GPU).
    // coarse_cut: first level every GPU holds and solves whole (0 = the first with <= 128 planes).
    // computeEF() differentiates each slab on its own GPU.
    MultiGridSolverHIP(World &world, int max_it, double tol, int depth, const std::vector<int> &devices,
                       int coarse_cut = 0);
    ~MultiGridSolverHIP() override;
    MultiGridSolverHIP(const MultiGridSolverHIP &) = delete;
    MultiGridSolverHIP &operator=(const MultiGridSolverHIP &) = delete;

    bool solve() override;
    // No `override`: the reference's base declares computeEF non-virtual (it is then hidden, not
    // overridden); against a base that declares it virtual this is an override all the same, which
    // clang remarks on.
#if defined(__clang__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winconsistent-missing-override"
#endif
    void computeEF();
#if defined(__clang__)
#pragma clang diagnostic pop
#endif

    enum ErrorPolicy { Throw, ReturnFalse };
    void setErrorPolicy(ErrorPolicy p) { on_error = p; }
    // on: the caller promises not to write world.phi between this solver's calls except through
    // invalidate(); the upload of phi is then skipped while the faces and a sample of the interior
    // still hold the last result.  Off (the default): every call uploads world.phi.
    void setAssumePhiUntouched(bool on) { assume_untouched = on; if (!on) device_phi_valid = false; }
    // world.phi was edited by someone else: upload it again at the next solve() / computeEF()
    void invalidate() { device_phi_valid = false; }

    // diagnostics of the last solve()
    unsigned cycles() const { return last_cycles; }
    double residualNorm() const { return last_l2; }
    double deviceMilliseconds() const;
    unsigned phiUploads() const { return phi_uploads; }  // how often world.phi went to the device

protected:
    // iteration loop run on the device between the upload and the download of solve()
    virtual int iterate(int *converged);

    unsigned max_solver_it;  // same names as the reference members (cpp/PotentialSolver.h:85-86)
    double tolerance;
    mg_handle *handle = nullptr;
    unsigned last_cycles = 0;
    double last_l2 = 0;

private:
    void init(int depth, int device, bool variant_b, const std::vector<int> &devices, int coarse_cut);
    const double *pack(Field &f);
    bool device_phi_current();
    void remember_phi();
    void upload_phi_if_stale();
    void download_ef();
    double *stage = nullptr;         // page-locked dense image of a field whose rows are not contiguous
    std::size_t stage_doubles = 0;
    std::vector<void *> registered;  // host blocks this solver page-locked in place
    std::vector<double> phi_sample;  // world.phi at the sample nodes when the device copy was last in step
    std::vector<double> phi_faces;   // and on every node of the six faces
    bool assume_untouched = false;
    bool device_phi_valid = false;
    unsigned phi_uploads = 0;
    ErrorPolicy on_error = Throw;
};

// Red-black SOR sweeps with the reference's stopping rule (norm of the interior residual tested
// every 25th sweep); a handle without coarse grids.
class GaussSeidelSolverHIP : public MultiGridSolverHIP {
public:
    GaussSeidelSolverHIP(World &world, int max_it, double tol, int device = 0)
        : MultiGridSolverHIP(world, max_it, tol, 0, device) {}

protected:
    int iterate(int *converged) override;
};

namespace hip {
struct GaussSeidelSolver : GaussSeidelSolverHIP { GaussSeidelSolver(World &w, int it, double tol) : GaussSeidelSolverHIP(w, it, tol) {} };
struct MultiGridSolver : MultiGridSolverHIP { MultiGridSolver(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 1) {} };
struct MultiGridSolverV2 : MultiGridSolverHIP { MultiGridSolverV2(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 2) {} };
struct MultiGridSolverV3 : MultiGridSolverHIP { MultiGridSolverV3(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 3) {} };
struct MultiGridSolverV4 : MultiGridSolverHIP { MultiGridSolverV4(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 4) {} };
struct MultiGridSolverV5 : MultiGridSolverHIP { MultiGridSolverV5(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 5) {} };
struct MultiGridSolverB : MultiGridSolverHIP { MultiGridSolverB(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 1, 0, true) {} };
struct MultiGridSolverV2B : MultiGridSolverHIP { MultiGridSolverV2B(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 2, 0, true) {} };
struct MultiGridSolverV3B : MultiGridSolverHIP { MultiGridSolverV3B(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 3, 0, true) {} };
struct MultiGridSolverV4B : MultiGridSolverHIP { MultiGridSolverV4B(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 4, 0, true) {} };
struct MultiGridSolverV5B : MultiGridSolverHIP { MultiGridSolverV5B(World &w, int it, double tol) : MultiGridSolverHIP(w, it, tol, 5, 0, true) {} };
}  // namespace hip

#endif
