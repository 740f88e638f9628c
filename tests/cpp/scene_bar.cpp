// A small scene through the class API the way ForceBuilder builds one
// (src/ForceBuilder.cpp:291-442): HyperElasticTet per tet in mesh order, then
// StaticAnchors, a MovingAnchor with a scripted ControlPoint (poordillo-style,
// samples/poordillo/poordillo.cpp:153-166), gravity, a pre-step callback.
// usage: scene_bar <in.bin> <out.bin> <frames> <iters>
//   in.bin : int32 n_nodes, n_tets, n_anchor, type(0 nh / 1 stvk); f64 x[3n], m[3n]; int32 tets[4*n_tets]; int32 anchors[n_anchor]; int32 moving_node
//   out.bin: f64 x[3n] after every frame, then global_idx / weight of every force as f64
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
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 1;
    const int n = hdr[0], nt = hdr[1], na = hdr[2], type = hdr[3];
    std::vector<double> x(3 * n), m(3 * n); std::vector<int> tets(4 * nt), anchors(na); int moving = -1;
    if (fread(x.data(), 8, 3 * n, f) != (size_t)3 * n || fread(m.data(), 8, 3 * n, f) != (size_t)3 * n) return 1;
    if (fread(tets.data(), 4, 4 * nt, f) != (size_t)4 * nt || fread(anchors.data(), 4, na, f) != (size_t)na || fread(&moving, 4, 1, f) != 1) return 1;
    fclose(f);
    const int frames = atoi(argv[3]), iters = atoi(argv[4]);

    System system;
    system.settings.verbose = 0; system.settings.timestep_s = 0.04; system.settings.admm_iters = iters;
    VectorXd xv(3 * n), mv(3 * n);
    for (int i = 0; i < 3 * n; ++i) { xv[i] = x[i]; mv[i] = m[i]; }
    system.add_nodes(xv, mv);
    for (int e = 0; e < nt; ++e)
        system.forces.push_back(std::shared_ptr<Force>(new HyperElasticTet(tets[4 * e], tets[4 * e + 1], tets[4 * e + 2], tets[4 * e + 3], 1e5, 1e5, 5, type ? "stvk" : "nh")));
    for (int a = 0; a < na; ++a) system.forces.push_back(std::shared_ptr<Force>(new StaticAnchor(anchors[a])));
    std::shared_ptr<ControlPoint> cp(new ControlPoint(Vector3d(x[3 * moving], x[3 * moving + 1], x[3 * moving + 2])));
    const Vector3d start = cp->pos, end = cp->pos + Vector3d(0.0, 0.05, 0.0);
    system.forces.push_back(std::shared_ptr<Force>(new MovingAnchor(moving, cp)));
    system.explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new ExplicitForce(Vector3d(0, -9.8, 0))));
    system.pre_step_callbacks.push_back([cp, start, end](System *s) { cp->pos = helper::smooth_move(s->elapsed_s, 0.0, 0.2, start, end); });
    if (!system.initialize()) return 2;

    FILE *o = fopen(argv[2], "wb");
    for (int fr = 0; fr < frames; ++fr) {
        if (fr == 3) cp->active = false;   // release: the control point now follows the node
        if (!system.step()) return 3;
        fwrite(system.m_x.data(), 8, 3 * n, o);
    }
    for (size_t i = 0; i < system.forces.size(); ++i) { double v[2] = {(double)system.forces[i]->global_idx, system.forces[i]->weight}; fwrite(v, 8, 2, o); }
    double cpv[3] = {cp->pos[0], cp->pos[1], cp->pos[2]};
    fwrite(cpv, 8, 3, o);
    fclose(o);
    printf("ok %d nodes %d forces elapsed %.2f\n", n, (int)system.forces.size(), system.elapsed_s);
    return 0;
}
