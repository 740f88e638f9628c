// plinkopony-style scene through the class API (samples/plinkopony/plinkopony.cpp:53-96):
// LinearTetStrain body, one CollisionForce over all nodes built from CollisionCylinder /
// CollisionSphere / CollisionFloor shapes, gravity, plus a WindForce-free ExplicitForce on a subset.
// usage: scene_plinko <in.bin> <out.bin> <frames> <iters>
//   in.bin: int32 n_nodes, n_tets, n_shapes; f64 x[3n], m[3n]; int32 tets[4 nt]; int32 types[ns]; f64 params[4 ns]
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "admm/System.hpp"
using namespace admm;

int main(int argc, char **argv) {
    if (argc < 5) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 1;
    const int n = hdr[0], nt = hdr[1], ns = hdr[2];
    std::vector<double> x(3 * n), m(3 * n), par(4 * ns); std::vector<int> tets(4 * nt), types(ns);
    if (fread(x.data(), 8, 3 * n, f) != (size_t)3 * n || fread(m.data(), 8, 3 * n, f) != (size_t)3 * n || fread(tets.data(), 4, 4 * nt, f) != (size_t)4 * nt) return 1;
    if (fread(types.data(), 4, ns, f) != (size_t)ns || fread(par.data(), 8, 4 * ns, f) != (size_t)4 * ns) return 1;
    fclose(f);
    System system;
    system.settings.verbose = 0; system.settings.timestep_s = 0.02; system.settings.admm_iters = atoi(argv[4]);
    VectorXd xv(3 * n), mv(3 * n);
    for (int i = 0; i < 3 * n; ++i) { xv[i] = x[i]; mv[i] = m[i]; }
    system.add_nodes(xv, mv);
    for (int e = 0; e < nt; ++e) system.forces.push_back(std::shared_ptr<Force>(new LinearTetStrain(tets[4 * e], tets[4 * e + 1], tets[4 * e + 2], tets[4 * e + 3], 1000.0)));
    std::vector<std::shared_ptr<CollisionShape> > shapes;
    for (int s = 0; s < ns; ++s) {
        const Vector3d c(par[4 * s], par[4 * s + 1], par[4 * s + 2]);
        if (types[s] == ADMM_SHAPE_FLOOR) shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionFloor(c)));
        else if (types[s] == ADMM_SHAPE_SPHERE) shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionSphere(c, par[4 * s + 3])));
        else shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionCylinder(c, Vector3d(1, 1, 1), par[4 * s + 3])));
    }
    system.forces.push_back(std::shared_ptr<Force>(new CollisionForce(shapes)));
    system.explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new ExplicitForce(Vector3d(0, -9.8, 0))));
    if (!system.initialize()) return 2;
    FILE *o = fopen(argv[2], "wb");
    for (int fr = 0; fr < atoi(argv[3]); ++fr) { if (!system.step()) return 3; fwrite(system.m_x.data(), 8, 3 * n, o); }
    fclose(o);
    printf("ok collision global_idx %d weight %g\n", system.forces.back()->global_idx, system.forces.back()->weight);
    return 0;
}
