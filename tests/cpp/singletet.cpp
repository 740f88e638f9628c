// The reference's "singletet" scenario (deps/admm-elastic-sca/samples/singletet.cpp:27-111)
// written against the host-side mirror (admm-elastic-sca_amd/host/admm/*.hpp):
// one LinearTetStrain tet, three StaticAnchors, dt = 1, node 4 displaced to
// x = 200, one step of 20 ADMM iterations.  The reference prints
// "Node 4 x: 171.571".
#include <cstdio>
#include <memory>
#include "admm/System.hpp"
using namespace admm;

int main() {
    System system;
    system.settings.verbose = 0;
    VectorXd x(4 * 3), m(4 * 3);
    m.fill(1); x.fill(0);
    x[0 * 3 + 1] = 1; x[2 * 3 + 2] = 1; x[3 * 3 + 0] = 1;
    system.add_nodes(x, m);
    for (int i = 0; i < 3; ++i) system.forces.push_back(std::shared_ptr<Force>(new StaticAnchor(i)));
    system.forces.push_back(std::shared_ptr<LinearTetStrain>(new LinearTetStrain(0, 1, 2, 3, 1.0)));
    system.settings.timestep_s = 1.0;
    if (!system.initialize()) return 2;
    system.m_x[3 * 3] = 200.0;
    system.settings.admm_iters = 20;
    if (!system.step()) return 3;
    printf("\n======\nSolver: ADMM, Max Iters: 20, Tet Force: Linear\nNode 4 x: %.6g\n======\n", system.m_x[3 * 3]);
    printf("full: %.17g %.17g %.17g\n", system.m_x[9], system.m_x[10], system.m_x[11]);
    return 0;
}
