#include "Output.h"

#include <fstream>
#include <iostream>

namespace {

// one k-plane per line, i fastest within it (cpp/Field.h:221-228)
template <typename T>
void write_nodes(std::ostream &out, Field_<T> &f)
{
    for (int k = 0; k < f.nk; k++) {
        for (int j = 0; j < f.nj; j++)
            for (int i = 0; i < f.ni; i++) out << f[i][j][k] << " ";
        out << "\n";
    }
}

template <typename T>
void data_array(std::ostream &out, const char *name, int components, Field_<T> &f)
{
    out << "<DataArray Name=\"" << name << "\" NumberOfComponents=\"" << components
        << "\" format=\"ascii\" type=\"Float64\">\n";
    write_nodes(out, f);
    out << "</DataArray>\n";
}

}  // namespace

bool Output::fields(World &world, const std::string &path)
{
    std::ofstream out(path);
    if (!out.is_open()) {
        std::cerr << "Could not open " << path << std::endl;
        return false;
    }
    double3 x0 = world.getX0(), dh = world.getDh();
    out << "<VTKFile type=\"ImageData\">\n";
    out << "<ImageData Origin=\"" << x0[0] << " " << x0[1] << " " << x0[2] << "\" ";
    out << "Spacing=\"" << dh[0] << " " << dh[1] << " " << dh[2] << "\" ";
    out << "WholeExtent=\"0 " << world.ni - 1 << " 0 " << world.nj - 1 << " 0 " << world.nk - 1 << "\">\n";
    out << "<PointData>\n";
    data_array(out, "NodeVol", 1, world.node_vol);
    data_array(out, "phi", 1, world.phi);
    data_array(out, "rho", 1, world.rho);
    data_array(out, "ef", 3, world.ef);
    out << "</PointData>\n";
    out << "</ImageData>\n";
    out << "</VTKFile>\n";
    return bool(out);
}
