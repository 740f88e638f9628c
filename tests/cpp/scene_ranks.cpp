// One admm::System per rank through the class API (System::shard): the scene of scene_bar.cpp -- HyperElasticTet bar or a
// cloth-like spring/anchor set-up is not needed here: NH / StVK tets, StaticAnchors, a scripted MovingAnchor that is released
// at frame 3, gravity -- run by `world` PROCESSES that each build the whole System and own a shard of the elements.
// Transport: comm::ShmAllReduce (host memory, one node), so that the ranks can share one GPU in the tests; on a multi-GPU
// node leave shm_name empty and the ranks meet through RCCL (System::shard.rccl_id_file, from the launcher's environment).
// usage: scene_ranks <in.bin> <out.bin> <frames> <iters> <rank> <world> <mode 0 contiguous / 1 subtree> [shm_name]
//   in.bin as scene_bar; out.bin: f64 x[3n] after every frame, then the control point (3)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "admm/System.hpp"
using namespace admm;

int main(int argc, char **argv) {
    if (argc < 8) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 1;
    const int n = hdr[0], nt = hdr[1], na = hdr[2], type = hdr[3];
    std::vector<double> x(3 * n), m(3 * n); std::vector<int> tets(4 * nt), anchors(na); int moving = -1;
    if (fread(x.data(), 8, 3 * n, f) != (size_t)3 * n || fread(m.data(), 8, 3 * n, f) != (size_t)3 * n) return 1;
    if (fread(tets.data(), 4, 4 * nt, f) != (size_t)4 * nt || fread(anchors.data(), 4, na, f) != (size_t)na || fread(&moving, 4, 1, f) != 1) return 1;
    fclose(f);
    const int frames = atoi(argv[3]), iters = atoi(argv[4]), rank = atoi(argv[5]), world = atoi(argv[6]), mode = atoi(argv[7]);

    System system;
    system.settings.verbose = 0; system.settings.timestep_s = 0.04; system.settings.admm_iters = iters;
    comm::ShmAllReduce shm;
    if (world > 1) {
        system.shard.rank = rank; system.shard.world = world; system.shard.mode = mode ? ADMM_SHARD_SUBTREE : ADMM_SHARD_CONTIGUOUS;
        if (argc > 8 && argv[8][0]) {
            if (!shm.open(argv[8], rank, world, 1 << 16, 60.0)) { fprintf(stderr, "rank %d: shared-memory segment %s unavailable\n", rank, argv[8]); return 4; }
            system.shard.host_allreduce = &comm::ShmAllReduce::hook; system.shard.host_allreduce_user = &shm;
        } else {
            const int local = system.shard.from_env();       // RCCL: id file from the launcher's environment
            if (local >= 0) system.device_id = local;
        }
    }
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
    double cpv[3] = {cp->pos[0], cp->pos[1], cp->pos[2]};
    fwrite(cpv, 8, 3, o);
    fclose(o);
    printf("ok rank %d of %d: %d nodes %d forces elapsed %.2f\n", rank, world, n, (int)system.forces.size(), system.elapsed_s);
    return 0;
}
