// Computational domain with the reference's surface (cpp/World.h:38-116): node counts,
// extents -> spacing, and the fields the potential solvers read and write.  Particle,
// time-stepping and diagnostic members of the reference are outside the solver path and
// are not provided.
#ifndef MG_HOST_WORLD_H
#define MG_HOST_WORLD_H

#include "Field.h"

namespace Const {
const double EPS_0 = 8.85418782e-12;  // C/(V*m), vacuum permittivity (cpp/World.h:14)
}

class World {
public:
    World(int ni, int nj, int nk);

    // mesh origin and opposite corner; dh = (xm - x0)/(n - 1) per axis (cpp/World.cpp:23-37)
    void setExtents(const double3 x0, const double3 xm);
    double3 getX0() const { return x0; }
    double3 getXm() const { return xm; }
    double3 getXc() const { return xc; }
    double3 getDh() const { return dh; }

    // 0 V Dirichlet walls: phi = 0 and object_id = 1 on the six faces (cpp/World.cpp:40-52)
    void setBoundaries();

    const int ni, nj, nk;
    const int3 nn;

    // 0.5 * EPS_0 * sum(|E|^2 * node volume) (cpp/World.cpp:88-100)
    double getPE();

    Field phi;         // potential; Dirichlet values live in its face nodes
    Field rho;         // charge density
    Field node_vol;    // dh[0]*dh[1]*dh[2], halved per wall the node lies on (cpp/World.cpp:74-85)
    Field3 ef;         // electric field
    FieldI object_id;  // 1 on fixed (Dirichlet) nodes

protected:
    void computeNodeVolumes();
    double3 x0, dh, xm, xc;
};

#endif
