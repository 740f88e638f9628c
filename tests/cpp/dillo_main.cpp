// BASELINE.json configs[2] / SURVEY 8(d) config 3: the shipped poordillo scene with the sample's grabbers
// (samples/poordillo/poordillo.cpp:133-166: MovingAnchors on the vertices within 0.2 of (.6,.8,.5) and
// (-.25,-.6,-.1)), the mouse replaced by a script: helper::smooth_move drags the hand to +2 x and the foot to
// -2 x over t in [1, 3] s; optionally the hand is released (the sample's H key: active = false, weight = 0,
// recompute_weights()) at a given frame.
//
// The file uses nothing but the reference's public API (SimContext::load / initialize, scene->objects_map,
// system->forces / pre_step_callbacks / step / recompute_weights), so the SAME source is compiled
//   (a) with the real reference                                    -> golden trajectories   (oracle/Makefile dillo_ref)
//   (b) with the reference's own src/SimContext.cpp + src/ForceBuilder.cpp over the mirror admm-elastic-sca_amd/host/admm
//       and libadmm_hip.so ("existing scenes drop in unchanged")   -> oracle/_ref/dillo_hip  (oracle/Makefile hip_callers)
//   (c) with the headless loader admm-elastic-sca_amd/host/SimContext.hpp                   (tests, GPU box)
//
//   dillo_main <xml> <out.bin> <frames> [release_frame = -1] [perturb_ulps = 0]
// out.bin: int32 dof, n_hand, n_foot; then per frame f64 x[dof]; then f64 hand control point 0 (3) after the run.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "SimContext.hpp"

using namespace admm;

struct Grabber {
    trimesh::point c;
    double rad;
    std::vector<Eigen::Vector3d> start, end;
    std::vector<std::shared_ptr<ControlPoint> > points;
    std::vector<int> ids;
    void collect(const std::vector<trimesh::point> &verts) {
        for (size_t i = 0; i < verts.size(); ++i) {
            if (trimesh::len(verts[i] - c) < rad) {      // poordillo.cpp:37
                const Eigen::Vector3d p(verts[i][0], verts[i][1], verts[i][2]);
                ids.push_back((int)i); start.push_back(p);
                points.push_back(std::shared_ptr<ControlPoint>(new ControlPoint(p)));
            }
        }
    }
    void aim(double ex, double ey, double ez) {
        const Eigen::Vector3d disp(ex - c[0], ey - c[1], ez - c[2]);
        for (size_t i = 0; i < start.size(); ++i) end.push_back(start[i] + disp);
    }
    void update(double t, double t0, double t1) { for (size_t i = 0; i < points.size(); ++i) points[i]->pos = helper::smooth_move(t, t0, t1, start[i], end[i]); }
};

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: dillo_main <xml> <out.bin> <frames> [release_frame] [perturb_ulps]\n"); return 1; }
    const int frames = std::atoi(argv[3]);
    const int release = argc > 4 ? std::atoi(argv[4]) : -1;
    const int ulps = argc > 5 ? std::atoi(argv[5]) : 0;
    try {
        SimContext context;
        context.load(argv[1]);
        context.system->settings.verbose = 0;
        Grabber hand, foot;
        hand.c = trimesh::point(.6, .8, .5); hand.rad = 0.2;
        foot.c = trimesh::point(-.25, -.6, -.1); foot.rad = 0.2;
        const std::vector<trimesh::point> &verts = context.scene->objects_map["dillo"]->get_TriMesh()->vertices;
        hand.collect(verts); foot.collect(verts);
        for (size_t i = 0; i < hand.ids.size(); ++i) context.system->forces.push_back(std::shared_ptr<Force>(new MovingAnchor(hand.ids[i], hand.points[i])));
        for (size_t i = 0; i < foot.ids.size(); ++i) context.system->forces.push_back(std::shared_ptr<Force>(new MovingAnchor(foot.ids[i], foot.points[i])));
        hand.aim(2.6, .8, .5);
        foot.aim(-2.25, -.6, -.1);
        context.initialize();
        System &S = *context.system;
        if (ulps) { double &v = S.m_x[7]; for (int q = 0; q < ulps; ++q) v = std::nextafter(v, 1e9); }
        S.pre_step_callbacks.push_back([&](System *sys) { hand.update(sys->elapsed_s, 1.0, 3.0); foot.update(sys->elapsed_s, 1.0, 3.0); });
        FILE *f = std::fopen(argv[2], "wb");
        if (!f) return 1;
        const int hdr[3] = {(int)S.m_x.size(), (int)hand.ids.size(), (int)foot.ids.size()};
        std::fwrite(hdr, 4, 3, f);
        for (int fr = 0; fr < frames; ++fr) {
            if (fr == release) {     // poordillo.cpp:203-212 (key H)
                for (size_t i = 0; i < hand.points.size(); ++i) { hand.points[i]->active = false; hand.points[i]->anchorForce->weight = 0.f; }
                S.recompute_weights();
            }
            if (!S.step()) { std::fclose(f); return 3; }
            std::fwrite(S.m_x.data(), 8, S.m_x.size(), f);
        }
        const double cp[3] = {hand.points[0]->pos[0], hand.points[0]->pos[1], hand.points[0]->pos[2]};
        std::fwrite(cp, 8, 3, f);
        std::fclose(f);
        std::printf("dillo_main: %d dof, %zu hand + %zu foot anchors, %d frames, dt %g, %d iterations\n", hdr[0], hand.ids.size(), foot.ids.size(), frames, S.settings.timestep_s, S.settings.admm_iters);
    } catch (std::exception &e) {
        std::fprintf(stderr, "dillo_main: %s\n", e.what());
        return 2;
    }
    return 0;
}
