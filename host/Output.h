// Field output in the reference's on-disk format (cpp/Output.cpp:12-65): a VTK ImageData
// (.vti) file with ASCII point data, every array written in the logical node order of
// cpp/Field.h:221-228 -- i fastest, then j, one k-plane per text line.
#ifndef MG_HOST_OUTPUT_H
#define MG_HOST_OUTPUT_H

#include <string>

#include "World.h"

namespace Output {
// Writes NodeVol, phi, rho and ef of `world` to `path`, in the reference's order.  The reference's
// file additionally carries per-species number densities (between rho and ef), which belong to
// its particle side.  Returns false
// (after a message on stderr, like the reference) when the file cannot be opened.
bool fields(World &world, const std::string &path);
}  // namespace Output

#endif
