// User-written plug-ins against the reference's extension surface (Force.hpp:37-57, ExplicitForce.hpp:51-59,
// CollisionShape.hpp:27-45; "you can just push them on the forces vector", samples/singletet.cpp:100-102).
// This file compiles UNCHANGED against the real reference (tests/golden/make_golden_user.py builds it with the reference's
// own headers and sources to record the expected trajectories) and against the mirror in admm-elastic-sca_amd/host/admm.
// It only uses what both offer: operator[], size(), Eigen::Triplet<double>(row, col, value).
//
//   user_force <mode> <out.bin> <frames> <iters> [n]
//   mode 0  net of the built-in Spring                      (device kernel)
//   mode 1  the same net with MySpring, a user subclass doing Spring's arithmetic  -> must equal mode 0 bit for bit
//   mode 2  MySpring net + ShellForce (not in the reference: keeps a node on a sphere) + a user ExplicitForce (swirl)
//   mode 3  built-in springs + CollisionForce over {CollisionFloor, user-written SlabShape}
// Output: frames x 3n doubles (m_x after every frame), then the user forces' global_idx and weight.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <memory>
#include <vector>

#include "System.hpp"
#include "AnchorForce.hpp"
#include "CollisionFloor.hpp"
#include "CollisionForce.hpp"
#include "ExplicitForce.hpp"

using namespace admm;

// Spring's arithmetic (Force.cpp:29-71) written by a user
class MySpring : public Force {
public:
    MySpring(int a_, int b_, double k_) : a(a_), b(b_), k(k_), rest(0) {}
    void initialize(const Eigen::VectorXd &x, const Eigen::VectorXd &, const Eigen::VectorXd &, const double) {
        const double d0 = x[3 * a] - x[3 * b], d1 = x[3 * a + 1] - x[3 * b + 1], d2 = x[3 * a + 2] - x[3 * b + 2];
        rest = std::sqrt(d0 * d0 + (d1 * d1 + d2 * d2));     // Eigen's 3-vector reduction order, like disp.norm()
        weight = std::sqrt(k);
    }
    void get_selector(const Eigen::VectorXd &, std::vector<Eigen::Triplet<double> > &triplets, std::vector<double> &weights) {
        global_idx = (int)weights.size();
        for (int i = 0; i < 3; ++i) {
            triplets.push_back(Eigen::Triplet<double>(i + global_idx, 3 * a + i, 1.0));
            triplets.push_back(Eigen::Triplet<double>(i + global_idx, 3 * b + i, -1.0));
        }
        for (int i = 0; i < 3; ++i) weights.push_back(weight);
    }
    void project(double, const Eigen::VectorXd &Dx, Eigen::VectorXd &u, Eigen::VectorXd &z) const {
        double d[3];
        for (int i = 0; i < 3; ++i) d[i] = Dx[global_idx + i] + u[global_idx + i];
        const double nrm = std::sqrt(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2]));
        const double c = 1.0 / (weight * weight + k);
        for (int i = 0; i < 3; ++i) {
            double dn = d[i] / nrm;
            if (nrm <= 0.0) dn = 0.0;
            const double p = rest * dn;
            const double zi = c * (k * p + weight * weight * d[i]);
            u[global_idx + i] += (Dx[global_idx + i] - zi);
            z[global_idx + i] = zi;
        }
    }
    int a, b;
    double k, rest;
};

// Not in the reference: pulls node `idx` onto the sphere |x - c| = R
class ShellForce : public Force {
public:
    ShellForce(int idx_, double cx, double cy, double cz, double R_, double k_) : idx(idx_), R(R_), k(k_) { c[0] = cx; c[1] = cy; c[2] = cz; }
    void initialize(const Eigen::VectorXd &, const Eigen::VectorXd &, const Eigen::VectorXd &, const double) { weight = std::sqrt(k); }
    void get_selector(const Eigen::VectorXd &, std::vector<Eigen::Triplet<double> > &triplets, std::vector<double> &weights) {
        global_idx = (int)weights.size();
        for (int i = 0; i < 3; ++i) { triplets.push_back(Eigen::Triplet<double>(i + global_idx, 3 * idx + i, 1.0)); weights.push_back(weight); }
    }
    void project(double, const Eigen::VectorXd &Dx, Eigen::VectorXd &u, Eigen::VectorXd &z) const {
        double d[3], r[3];
        for (int i = 0; i < 3; ++i) { d[i] = Dx[global_idx + i] + u[global_idx + i]; r[i] = d[i] - c[i]; }
        const double nrm = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int i = 0; i < 3; ++i) {
            const double p = nrm > 0.0 ? c[i] + R * (r[i] / nrm) : d[i];
            const double zi = (k * p + weight * weight * d[i]) / (weight * weight + k);
            u[global_idx + i] += (Dx[global_idx + i] - zi);
            z[global_idx + i] = zi;
        }
    }
    int idx;
    double c[3], R, k;
};

// A user explicit force: acceleration field that swirls around the y axis
class SwirlForce : public ExplicitForce {
public:
    SwirlForce(double strength_) : strength(strength_) {}
    void project(double dt, Eigen::VectorXd &x, Eigen::VectorXd &v, Eigen::VectorXd &) const {
        for (int i = 0; i < (int)x.size() / 3; ++i) { v[3 * i] += dt * (-strength * x[3 * i + 2]); v[3 * i + 2] += dt * (strength * x[3 * i]); }
    }
    double strength;
};

// A user collision shape: the half space x > xmax is solid
class SlabShape : public CollisionShape {
public:
    SlabShape(double xmax_) : CollisionShape(Eigen::Vector3d(xmax_, 0, 0)), xmax(xmax_) {}
    double isColliding(Eigen::Vector3d pos) const { return pos[0] - xmax; }
    Eigen::Vector3d projectOut(const Eigen::Vector3d currPos) const { return Eigen::Vector3d(xmax, currPos[1], currPos[2]); }
    double xmax;
};

int main(int argc, char **argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: user_force mode out frames iters [n]\n"); return 1; }
    const int mode = std::atoi(argv[1]), frames = std::atoi(argv[3]), iters = std::atoi(argv[4]);
    const int n = argc > 5 ? std::atoi(argv[5]) : 9;
    System system;
    system.settings.verbose = 0;
    system.settings.timestep_s = 0.04;
    system.settings.admm_iters = iters;
    Eigen::VectorXd x(3 * n * n), m(3 * n * n);
    const double h = 0.1;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const int q = i + n * j;
        x[3 * q] = h * i; x[3 * q + 1] = 0.02 * std::sin(0.7 * i + 0.3 * j); x[3 * q + 2] = h * j;
        m[3 * q] = m[3 * q + 1] = m[3 * q + 2] = 0.05;
    }
    system.add_nodes(x, m);
    const double k = 400.0;
    std::vector<std::shared_ptr<Force> > mine;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const int q = i + n * j;
        const int nb[3] = {i + 1 < n ? q + 1 : -1, j + 1 < n ? q + n : -1, (i + 1 < n && j + 1 < n) ? q + n + 1 : -1};
        for (int t = 0; t < 3; ++t) if (nb[t] >= 0) {
            std::shared_ptr<Force> f;
            if (mode == 1 || mode == 2) { f.reset(new MySpring(q, nb[t], k)); mine.push_back(f); }
            else f.reset(new Spring(q, nb[t], k));
            system.forces.push_back(f);
        }
    }
    system.forces.push_back(std::shared_ptr<Force>(new StaticAnchor(0)));
    system.forces.push_back(std::shared_ptr<Force>(new StaticAnchor(n - 1)));
    if (mode == 2) {
        std::shared_ptr<Force> sh(new ShellForce(n * n - 1, h * (n - 1), -0.3, h * (n - 1), 0.25, 900.0));
        mine.push_back(sh);
        system.forces.push_back(sh);
        system.explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new SwirlForce(1.5)));
    }
    std::vector<std::shared_ptr<CollisionShape> > shapes;
    if (mode == 3) {
        shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionFloor(Eigen::Vector3d(0, -0.15, 0))));
        shapes.push_back(std::shared_ptr<CollisionShape>(new SlabShape(0.55)));
        std::shared_ptr<Force> cf(new CollisionForce(shapes));
        mine.push_back(cf);
        system.forces.push_back(cf);
    }
    system.explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new ExplicitForce(Eigen::Vector3d(0, -9.8, 0))));
    if (!system.initialize()) return 2;
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 4;
    const std::clock_t c0 = std::clock();
    struct timespec w0, w1;
    clock_gettime(CLOCK_MONOTONIC, &w0);
    for (int fr = 0; fr < frames; ++fr) {
        if (!system.step()) { std::fclose(f); return 3; }
        std::fwrite(system.m_x.data(), sizeof(double), 3 * n * n, f);
    }
    clock_gettime(CLOCK_MONOTONIC, &w1);
    (void)c0;
    const double wall = (w1.tv_sec - w0.tv_sec) + 1e-9 * (w1.tv_nsec - w0.tv_nsec);
    for (size_t i = 0; i < mine.size(); ++i) { const double gw[2] = {(double)mine[i]->global_idx, mine[i]->weight}; std::fwrite(gw, sizeof(double), 2, f); }
    std::fclose(f);
    std::printf("user_force: mode %d, %d nodes, %zu forces (%zu user), %d frames x %d iterations, %.1f us per ADMM iteration\n", mode, n * n, system.forces.size(), mine.size(), frames, iters,
                1e6 * wall / (frames * (double)iters));
    return 0;
}
