// The reference's "singlenode" scenario (deps/admm-elastic-sca/samples/singlenode.cpp:25-73):
// one free node, gravity -9.8f, dt = 1, four steps: y = -9.8, -29.4, -58.8, -98.
#include <cstdio>
#include <memory>
#include "admm/System.hpp"
using namespace admm;

int main() {
    System system;
    system.settings.verbose = 0;
    system.m_x.resize(3); system.m_x.fill(0);
    system.m_v.resize(3); system.m_v.fill(0);
    system.m_masses.resize(3); system.m_masses.fill(1);
    system.explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new ExplicitForce(Vector3d(0.f, -9.8f, 0.f))));
    system.settings.timestep_s = 1.0;
    if (!system.initialize()) return 2;
    system.settings.admm_iters = 20;
    for (int i = 0; i < 4; ++i) {
        if (!system.step()) return 3;
        printf("step: %d, pos: (%g, %g, %g)\n", i, system.m_x[0], system.m_x[1], system.m_x[2]);
    }
    return 0;
}
