// The plug-in boundary of the reference (cpp/PotentialSolver.h:49-68): an abstract solver
// bound to a World, used through a base pointer (cpp/Main.cpp:59,124,157).
#ifndef MG_HOST_POTENTIAL_SOLVER_H
#define MG_HOST_POTENTIAL_SOLVER_H

#include "World.h"

class PotentialSolver {
public:
    explicit PotentialSolver(World &w) : world(w) {}
    virtual ~PotentialSolver() {}
    virtual bool solve() = 0;
    // E = -grad(phi), central differences inside, one-sided 3-point on the faces
    // (cpp/PotentialSolver.cpp:418-458): host loop.  Virtual here (the reference's is not) so
    // that the device solvers can answer the driver's solver->computeEF() call
    // (cpp/Main.cpp:126,166) from the potential they still hold in device memory.
    // -DMG_REFERENCE_COMPUTE_EF gives the reference's non-virtual declaration
    // (cpp/PotentialSolver.h:63), to build the solvers the way the reference tree would.
#ifdef MG_REFERENCE_COMPUTE_EF
    void computeEF();
#else
    virtual void computeEF();
#endif

protected:
    World &world;
};

#endif
