// Minimal driver in the shape of the reference's main loop around the solver
// (cpp/Main.cpp:29-39, :88-100, :124-126, :155-166): cubic mesh of argv[1] nodes, the reference
// box, 0 V walls, a multigrid solver chosen by depth, solve() timed with the same clock.
// The particle species that produce rho in the reference are replaced by the deterministic
// synthetic charge density used by this repository's benchmarks.
//
//   box_mg [n=41] [depth=2] [tol=1e-4] [--b] [--dump file] [--dump-ef file]
//   --menu asks for the solver on stdin with the reference's two prompts (cpp/Main.cpp:62-107:
//   `printf '4\n2\n' | box_mg 41 --menu` is the reference's `printf '4\n2\n' | ./box 41`), with
//   the reference's limits (10000 iterations, tolerance 1e-4); of its menu, (1) Gauss-Seidel and
//   (4) MultiGrid V1-V5 are this library's solvers, the others are refused;
//   --b selects the reference's "B" solver variant (7-point restriction);
//   --vti file writes phi, rho and ef as VTK ImageData in the reference's ASCII format
//   --gpus 0,1,2,3 splits the grid into slabs over those GPUs (an ordinal may repeat); --cut L names
//   the first level every GPU holds whole (default: the first with at most 128 planes);
//   --pageable keeps the World's fields in ordinary memory (default: page-locked blocks);
//   --again repeats solve() twice more: once with world.phi untouched, once after the driver has
//   reset world.phi to zero; --assume-untouched lets the solver skip the upload of a world.phi
//   that still holds its last result (setAssumePhiUntouched: the first repeat then uploads nothing,
//   the second is noticed and uploaded);
//   --edit face|interior writes 5 V into one node of world.phi after the first solve -- an
//   off-centre node of the j = 0 wall (an electrode patch) or an interior node -- and solves again;
//   --invalidate announces that edit with invalidate(); --max-it N bounds every solve;
//   depth 0 selects the plain Gauss-Seidel/SOR solver; --dump writes phi as raw fp64 [i][j][k],
//   --dump-ef the electric field as raw fp64 [i][j][k][3]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <vector>

#include "../include/mg_hip.h"
#include "MultiGridSolverHIP.h"
#include "Output.h"

#ifndef MG_FIELD_ROW_PER_ALLOCATION
namespace {
// page-locked node blocks for the World's fields (see field_block_alloc in Field.h)
void *pinned_alloc(std::size_t bytes)
{
    void *p = nullptr;
    return mg_host_alloc(&p, bytes) == MG_OK ? p : nullptr;
}
void pinned_free(void *p) { mg_host_free(p); }
}  // namespace
#endif

int main(int argc, char **argv)
{
    int n = 41, depth = 2;
    double tol = 1e-4;
    const char *dump = nullptr, *dump_ef = nullptr, *vti = nullptr;
    bool variant_b = false, again = false, pageable = false, menu = false, assume_untouched = false, invalidate = false;
    const char *edit = nullptr;
    int max_it = 0, cut = 0;
    std::vector<int> gpus;
    int pos = 0;
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "--dump") && a + 1 < argc) dump = argv[++a];
        else if (!strcmp(argv[a], "--dump-ef") && a + 1 < argc) dump_ef = argv[++a];
        else if (!strcmp(argv[a], "--b")) variant_b = true;
        else if (!strcmp(argv[a], "--again")) again = true;
        else if (!strcmp(argv[a], "--menu")) menu = true;
        else if (!strcmp(argv[a], "--assume-untouched")) assume_untouched = true;
        else if (!strcmp(argv[a], "--invalidate")) invalidate = true;
        else if (!strcmp(argv[a], "--edit") && a + 1 < argc) edit = argv[++a];
        else if (!strcmp(argv[a], "--max-it") && a + 1 < argc) max_it = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--pageable")) pageable = true;
        else if (!strcmp(argv[a], "--cut") && a + 1 < argc) cut = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--gpus") && a + 1 < argc) {
            for (char *t = strtok(argv[++a], ","); t; t = strtok(nullptr, ",")) gpus.push_back(atoi(t));
        }
        else if (!strcmp(argv[a], "--vti") && a + 1 < argc) vti = argv[++a];
        else if (pos == 0) { n = atoi(argv[a]); pos++; }
        else if (pos == 1) { depth = atoi(argv[a]); pos++; }
        else if (pos == 2) { tol = atof(argv[a]); pos++; }
    }
    if (variant_b && gpus.size() > 1) {
        std::cerr << "box_mg: --b (7-point restriction) is not available together with --gpus" << std::endl;
        return 1;
    }
    if (edit && strcmp(edit, "face") && strcmp(edit, "interior")) {
        std::cerr << "box_mg: --edit takes face or interior" << std::endl;
        return 1;
    }
    std::cout << "Using mesh size : " << n << "x" << n << "x" << n << std::endl;
    if (menu) {
        int choice = 0, choice2 = 0;
        tol = 1e-4;
        std::cout << "Choose solver (1)Gauss-Seidel (2)FFT (3)Conjugate Gradient (4)MultiGrid (5) Preconditioned CG : ";
        std::cin >> choice;
        if (choice == 4) {
            std::cout << "MultiGrid V-Cycle size (1)V1 (2)V2 (3)V3 (4)V4 (5)V5: ";
            std::cin >> choice2;
            if (choice2 < 1 || choice2 > 5) {
                std::cout << "Invalid choice using V1 as default \n";
                choice2 = 1;
            }
            depth = choice2;
        } else if (choice == 2 || choice == 3 || choice == 5) {
            std::cerr << "box_mg: solver " << choice << " (FFT, CG, PCG) is not part of the multigrid library" << std::endl;
            return 1;
        } else {
            if (choice != 1) std::cout << "Invalid choice using Gauss-Seidel as default \n";
            depth = 0;
        }
    }
    try {
#ifndef MG_FIELD_ROW_PER_ALLOCATION
        if (!pageable) {
            field_block_alloc = pinned_alloc;
            field_block_free = pinned_free;
        }
#endif
        World world(n, n, n);
        world.setExtents({-0.2, -0.2, -0.1}, {0.2, 0.2, 0.3});
        world.setBoundaries();
        const double pi = std::acos(-1.0);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                for (int k = 0; k < n; k++) {
                    const double x = double(i) / (n - 1), y = double(j) / (n - 1), z = double(k) / (n - 1);
                    world.rho[i][j][k] = Const::EPS_0 * 1000.0 *
                                         (std::sin(pi * x) * std::sin(pi * y) * std::sin(pi * z) +
                                          0.5 * std::sin(5 * pi * x) * std::sin(3 * pi * y) * std::sin(7 * pi * z));
                }
        const int its = max_it > 0 ? max_it : (depth == 0 && !menu ? 100000 : 10000);
        MultiGridSolverHIP *mg = depth == 0       ? new GaussSeidelSolverHIP(world, its, tol)
                                 : gpus.size() > 1 ? new MultiGridSolverHIP(world, its, tol, depth, gpus, cut)
                                                   : new MultiGridSolverHIP(world, its, tol, depth, 0, variant_b);
        mg->setAssumePhiUntouched(assume_untouched);
        PotentialSolver *solver = mg;  // used through the base pointer, like the reference driver
        auto t0 = std::chrono::high_resolution_clock::now();
        const bool ok = solver->solve();
        auto t1 = std::chrono::high_resolution_clock::now();
        solver->computeEF();
        std::chrono::duration<double> dt = t1 - t0;
        std::cout << "Simulation took " << dt.count() << " secs for solver1 to solve potential" << std::endl;
        (void)pageable;
        std::cout << "converged=" << ok << " cycles=" << mg->cycles() << " L2=" << mg->residualNorm()
                  << " device_ms=" << mg->deviceMilliseconds() << std::endl;
        if (edit) {
            // a Dirichlet value written onto a wall away from its centre, or one interior node
            if (!strcmp(edit, "face")) world.phi[n / 3][0][n / 4 + 1] = 5.0;
            else world.phi[n / 3][n / 2 + 1][n / 4 + 1] = 5.0;
            if (invalidate) mg->invalidate();
            const bool ok2 = solver->solve();
            solver->computeEF();
            std::cout << "edit: converged=" << ok2 << " cycles=" << mg->cycles() << " phi_uploads=" << mg->phiUploads()
                      << " L2=" << mg->residualNorm() << std::endl;
        }
        if (again) {
            const bool ok2 = solver->solve();  // phi untouched: stops at its first test
            std::cout << "again: converged=" << ok2 << " cycles=" << mg->cycles() << " phi_uploads=" << mg->phiUploads()
                      << std::endl;
            world.phi.clear();                 // a new initial guess: must be noticed and uploaded
            const bool ok3 = solver->solve();
            solver->computeEF();
            std::cout << "reset: converged=" << ok3 << " cycles=" << mg->cycles() << " phi_uploads=" << mg->phiUploads()
                      << std::endl;
        }
        if (dump) {
            FILE *f = fopen(dump, "wb");
            bool good = f != nullptr;
            for (int i = 0; good && i < n; i++)
                for (int j = 0; good && j < n; j++) good = fwrite(world.phi[i][j], sizeof(double), n, f) == size_t(n);
            if (!good) {
                std::cerr << "cannot write " << dump << std::endl;
                return 3;
            }
            fclose(f);
        }
        if (dump_ef) {
            FILE *f = fopen(dump_ef, "wb");
            bool good = f != nullptr;
            for (int i = 0; good && i < n; i++)
                for (int j = 0; good && j < n; j++) good = fwrite(world.ef[i][j], sizeof(double3), n, f) == size_t(n);
            if (!good) {
                std::cerr << "cannot write " << dump_ef << std::endl;
                return 3;
            }
            fclose(f);
        }
        if (vti && !Output::fields(world, vti)) return 3;
        delete solver;
        return ok || edit ? 0 : 2;
    } catch (const std::exception &e) {
        std::cerr << "box_mg: " << e.what() << std::endl;
        return 1;
    }
}
