// force_init.hpp -- host-side Force::initialize() for every accelerated kind:
// rest-shape matrices, measures and weights, evaluated in the same operation
// order as the reference's Eigen expressions so that D and W are bit-identical
// (tests/test_assembly.py compares them with the compiled reference).
//
// Reference (deps/admm-elastic-sca/src/system/):
//   helper::init_tet_force            TetForce.cpp:28-57
//   LinearTetStrain/TetVolume/HyperElasticTet::initialize  TetForce.cpp:112-117,160-163,303-310
//   LimitedTriangleStrain::initialize TriangleForce.cpp:29-63 (TriArea inherits it); FungTriangle::initialize :171-210
//   BendForce::initialize             BendForce.cpp:26-56
//   Spring::initialize                Force.cpp:29-38
//   StaticAnchor ctor / initialize    AnchorForce.hpp:57-60, AnchorForce.cpp:31-35
#pragma once
#include <cmath>
#include "../../include/admm_kinds.h"

namespace admm_host {

struct V3d { double x, y, z; };
inline V3d sub(const V3d &a, const V3d &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// Eigen fixed-size reductions associate as a0 + (a1 + a2)
inline double dot(const V3d &a, const V3d &b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline double norm(const V3d &a) { return std::sqrt(dot(a, a)); }
inline V3d cross(const V3d &a, const V3d &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3d node(const double *x, int i) { return {x[3 * (size_t)i], x[3 * (size_t)i + 1], x[3 * (size_t)i + 2]}; }

// rest[12]: B (4x3, column-major), returns volume
inline double tet_rest(const int *idx, const double *x, double *B) {
    const V3d v0 = node(x, idx[0]), v1 = node(x, idx[1]), v2 = node(x, idx[2]), v3 = node(x, idx[3]);
    const V3d e0 = sub(v1, v0), e1 = sub(v2, v0), e2 = sub(v3, v0);
    // E = [e0 e1 e2] (columns); adjugate-based inverse as Eigen's 3x3 path does it
    const double m[3][3] = {{e0.x, e1.x, e2.x}, {e0.y, e1.y, e2.y}, {e0.z, e1.z, e2.z}}; // m[row][col]
    auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]; };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = c00 * m[0][0] + (c10 * m[1][0] + c20 * m[2][0]);
    const double invdet = 1.0 / det;
    double inv[3][3];
    inv[0][0] = c00 * invdet; inv[0][1] = c10 * invdet; inv[0][2] = c20 * invdet;
    for (int j = 0; j < 3; ++j) { inv[1][j] = cof(j, 1) * invdet; inv[2][j] = cof(j, 2) * invdet; }
    for (int j = 0; j < 3; ++j) {
        B[0 + 4 * j] = (-1.0 * inv[0][j] + -1.0 * inv[1][j]) + -1.0 * inv[2][j];
        B[1 + 4 * j] = (1.0 * inv[0][j] + 0.0 * inv[1][j]) + 0.0 * inv[2][j];
        B[2 + 4 * j] = (0.0 * inv[0][j] + 1.0 * inv[1][j]) + 0.0 * inv[2][j];
        B[3 + 4 * j] = (0.0 * inv[0][j] + 0.0 * inv[1][j]) + 1.0 * inv[2][j];
    }
    return std::fabs(dot(sub(v0, v3), cross(sub(v1, v3), sub(v2, v3)))) / 6.0;
}

// rest[6]: B (3x2, column-major), returns area
inline double tri_rest(const int *idx, const double *x, double *B) {
    const V3d x1 = node(x, idx[0]), x2 = node(x, idx[1]), x3 = node(x, idx[2]);
    const V3d e12 = sub(x2, x1), e13 = sub(x3, x1);
    double l = norm(e12);
    const V3d n1 = {e12.x / l, e12.y / l, e12.z / l};
    const double d = dot(e13, n1);
    const V3d t = {e13.x - d * n1.x, e13.y - d * n1.y, e13.z - d * n1.z};
    l = norm(t);
    const V3d n2 = {t.x / l, t.y / l, t.z / l};
    auto dseq = [](const V3d &a, const V3d &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }; // product coefficient order
    const double g00 = dseq(n1, e12), g10 = dseq(n2, e12), g01 = dseq(n1, e13), g11 = dseq(n2, e13);
    const double det = g00 * g11 - g10 * g01;
    const double invdet = 1.0 / det;
    const double i00 = g11 * invdet, i10 = -g10 * invdet, i01 = -g01 * invdet, i11 = g00 * invdet;
    const double Xi[2][2] = {{i00, i01}, {i10, i11}}; // Xi[row][col]
    for (int j = 0; j < 2; ++j) {
        B[0 + 3 * j] = -1.0 * Xi[0][j] + -1.0 * Xi[1][j];
        B[1 + 3 * j] = 1.0 * Xi[0][j] + 0.0 * Xi[1][j];
        B[2 + 3 * j] = 0.0 * Xi[0][j] + 1.0 * Xi[1][j];
    }
    return std::fabs(det / 2.0f);
}

inline void bend_rest(const int *idx, const double *x, double *alpha) {
    const V3d x0 = node(x, idx[0]), x1 = node(x, idx[1]), x2 = node(x, idx[2]), x3 = node(x, idx[3]);
    const V3d xA = sub(x0, x2), xB = sub(x1, x2), xC = {0, 0, 0}, xD = sub(x3, x2);
    const double area1 = 0.5 * norm(cross(xA, xD));
    const double area2 = 0.5 * norm(cross(xD, xB));
    const double hA = 2.0 * area1 / norm(xD);
    const double hB = 2.0 * area2 / norm(xD);
    const V3d nC = cross(sub(xC, xB), sub(xC, xA));
    const V3d nD = cross(sub(xD, xA), sub(xD, xB));
    alpha[0] = hB / (hA + hB);
    alpha[1] = hA / (hA + hB);
    alpha[2] = -norm(nD) / (norm(nC) + norm(nD));
    alpha[3] = -norm(nC) / (norm(nC) + norm(nD));
}

// Fills weight and rest[12] for one element; returns false for an unknown kind.
inline bool force_initialize(int kind, const int *idx, const double *params, const double *x, double *weight, double *rest) {
    for (int i = 0; i < 12; ++i) rest[i] = 0.0;
    switch (kind) {
    case ADMM_KIND_ANCHOR:
        *weight = params[0] > 0.0 ? params[0] : (double)1000.f;
        return true;
    case ADMM_KIND_COLLISION:   // CollisionForce.hpp:33: weight = use_weight
        *weight = params[0];
        return true;
    case ADMM_KIND_SPRING: {
        rest[0] = norm(sub(node(x, idx[0]), node(x, idx[1])));
        *weight = std::sqrt(params[0]);
        return true;
    }
    case ADMM_KIND_TET_LINEAR: case ADMM_KIND_TET_VOLUME: {
        double B[12]; const double vol = tet_rest(idx, x, B);
        for (int i = 0; i < 12; ++i) rest[i] = B[i];
        *weight = sqrtf((float)params[0]) * sqrtf((float)vol);
        return true;
    }
    case ADMM_KIND_TET_NH: case ADMM_KIND_TET_STVK: {
        double B[12]; const double vol = tet_rest(idx, x, B);
        for (int i = 0; i < 12; ++i) rest[i] = B[i];
        const double stiff = params[1] < params[0] ? params[1] : params[0]; // std::min(mu, lambda)
        *weight = sqrtf((float)stiff) * sqrtf((float)vol);
        return true;
    }
    case ADMM_KIND_TRI_STRAIN: case ADMM_KIND_TRI_AREA: {   // TriArea inherits LimitedTriangleStrain::initialize
        double B[6]; const double area = tri_rest(idx, x, B);
        for (int i = 0; i < 6; ++i) rest[i] = B[i];
        *weight = sqrtf((float)params[0]) * sqrtf((float)area);
        return true;
    }
    case ADMM_KIND_TRI_FUNG: {    // FungTriangle::initialize, TriangleForce.cpp:171-210: double sqrt
        double B[6]; const double area = tri_rest(idx, x, B);
        for (int i = 0; i < 6; ++i) rest[i] = B[i];
        *weight = std::sqrt(params[0]) * std::sqrt(area);
        return true;
    }
    case ADMM_KIND_BEND:
        bend_rest(idx, x, rest);
        *weight = std::sqrt(params[0]);
        return true;
    }
    return false;
}

// measure (volume / area) is needed by the blended kinds: k = stiffness * measure
inline double force_measure(int kind, const int *idx, const double *x) {
    double tmp[12];
    if (kind == ADMM_KIND_TET_LINEAR || kind == ADMM_KIND_TET_VOLUME || kind == ADMM_KIND_TET_NH || kind == ADMM_KIND_TET_STVK) return tet_rest(idx, x, tmp);
    if (kind == ADMM_KIND_TRI_STRAIN || kind == ADMM_KIND_TRI_AREA || kind == ADMM_KIND_TRI_FUNG) return tri_rest(idx, x, tmp);
    return 0.0;
}

} // namespace admm_host
