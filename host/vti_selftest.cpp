// Writes a small World with known node values through Output::fields (no GPU needed): the
// test-suite parses the file and checks header, node order and values.
#include <cstdio>

#include "Output.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: vti_selftest out.vti\n");
        return 2;
    }
    World world(3, 4, 5);
    world.setExtents({-0.2, -0.2, -0.1}, {0.2, 0.1, 0.3});
    for (int i = 0; i < world.ni; i++)
        for (int j = 0; j < world.nj; j++)
            for (int k = 0; k < world.nk; k++) {
                world.phi[i][j][k] = 100 * i + 10 * j + k + 0.5;
                world.rho[i][j][k] = -(100 * i + 10 * j + k);
                world.ef[i][j][k] = double3(i, j, k);
            }
    return Output::fields(world, argv[1]) ? 0 : 1;
}
