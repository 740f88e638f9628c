// Scene.hpp -- headless restatement of the parts of mclscene / trimesh2 that the
// reference's scene ingest (src/SimContext.cpp, src/ForceBuilder.{hpp,cpp}) goes
// through, so the shipped XML scenes can be loaded on a box with no GL stack:
//
//   XML text -> Component / Param lists          (deps/mclscene/include/MCL/Param.hpp:122-201,
//                                                 deps/mclscene/src/SceneManager.cpp:37-147)
//   scale / translate / rotate -> xform          (Param.hpp:132-160: the matrix is written to TEXT with
//                                                 the stream's default 6 significant digits and parsed
//                                                 back, so a 20 degree rotation carries 6-digit cosines;
//                                                 trimesh2 include/XForm.h:122-143,311-330,474-484,487-523)
//   tetmesh: <File>.node/.ele -> float vertices  (deps/mclscene/src/TetMesh.cpp:133-228; surface 231-271)
//   plane: make_sym_plane                        (trimesh2 include/TriMeshBuilder.h:24-59,116-171)
//   across-edge face adjacency (bend hinges)     (trimesh2 libsrc/TriMesh_connectivity.cc:58-134)
//
// Everything the solver reads is reproduced bit for bit: vertices are `float`, the
// transform is applied in double and rounded back to float per coordinate, and the
// same libstdc++ stream conversions parse and print the numbers.  What is NOT here:
// rendering data (normals, tstrips, textures, materials, lights, cameras, BVH), trimesh2's
// readers for mesh files other than Wavefront OBJ, PLY and OFF (3ds, stl, sm ...) and point clouds: such
// objects are accepted as static scenery (their parameters are kept); giving one a <Force> is an error.
//   trimesh: <File> .obj / .ply / .off           (trimesh2 libsrc/TriMesh_io.cc:232-337,342-556,736-806,870-1150,1243-1407) -- round 5
//   sphere / box / beam / cylinder / torus       (mclscene DefaultBuilders.hpp:83-256 over trimesh2 TriMeshBuilder.h:220-556,
//                                                 libsrc/remove.cc) -- round 5: tessellated like the reference does, forces attach
//
// Bit parity with the reference needs the arithmetic below compiled without FMA
// contraction (-ffp-contract=off, the default x86-64 baseline has no FMA anyway).
#pragma once
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <strings.h>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace trimesh {

template <int D, class T> struct Vec {
    T v[D];
    Vec() { for (int i = 0; i < D; ++i) v[i] = T(0); }
    Vec(T a, T b) { v[0] = a; v[1] = b; }
    Vec(T a, T b, T c) { v[0] = a; v[1] = b; v[2] = c; }
    T &operator[](int i) { return v[i]; }
    const T &operator[](int i) const { return v[i]; }
    typedef T value_type;
};
// trimesh2 Vec.h: component-wise difference / sum, len2 accumulated front to back in T, len = sqrt(len2)
template <int D, class T> inline Vec<D, T> operator-(const Vec<D, T> &a, const Vec<D, T> &b) { Vec<D, T> r; for (int i = 0; i < D; ++i) r[i] = a[i] - b[i]; return r; }
template <int D, class T> inline Vec<D, T> operator+(const Vec<D, T> &a, const Vec<D, T> &b) { Vec<D, T> r; for (int i = 0; i < D; ++i) r[i] = a[i] + b[i]; return r; }
template <int D, class T> inline T len2(const Vec<D, T> &v) { T l2 = v[0] * v[0]; for (int i = 1; i < D; ++i) l2 += v[i] * v[i]; return l2; }
template <int D, class T> inline T len(const Vec<D, T> &v) { return std::sqrt(len2(v)); }
typedef Vec<3, float> vec;
typedef Vec<3, float> point;
typedef Vec<3, float> vec3;
typedef Vec<2, float> vec2;
typedef Vec<4, float> vec4;

// 4x4, column-major, double (trimesh2's `xform`)
struct xform {
    double m[16];
    xform() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
    double &operator[](int i) { return m[i]; }
    const double &operator[](int i) const { return m[i]; }
    static xform identity() { return xform(); }
    static xform trans(double tx, double ty, double tz) { xform r; r[12] = tx; r[13] = ty; r[14] = tz; return r; }
    static xform scale(double sx, double sy, double sz) { xform r; r[0] = sx; r[5] = sy; r[10] = sz; return r; }
    // angle in radians about (rx, ry, rz); XForm.h:126-143
    static xform rot(double angle, double rx, double ry, double rz) {
        const double l = std::sqrt(rx * rx + ry * ry + rz * rz);
        if (l == 0.0) return xform();
        const double l1 = 1.0 / l, x = rx * l1, y = ry * l1, z = rz * l1;
        const double s = std::sin(angle), c = std::cos(angle);
        const double xs = x * s, ys = y * s, zs = z * s, c1 = 1.0 - c;
        const double xx = c1 * x * x, yy = c1 * y * y, zz = c1 * z * z;
        const double xy = c1 * x * y, xz = c1 * x * z, yz = c1 * y * z;
        xform r;
        r[0] = xx + c;  r[1] = xy + zs; r[2] = xz - ys;  r[3] = 0;
        r[4] = xy - zs; r[5] = yy + c;  r[6] = yz + xs;  r[7] = 0;
        r[8] = xz + ys; r[9] = yz - xs; r[10] = zz + c;  r[11] = 0;
        r[12] = 0; r[13] = 0; r[14] = 0; r[15] = 1;
        return r;
    }
    template <class S> static xform rot(double angle, const S &axis) { return rot(angle, axis[0], axis[1], axis[2]); }
};

// XForm.h:311-330: each entry is a left-to-right sum of four products
static inline xform operator*(const xform &a, const xform &b) {
    xform r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[i + 4 * j] = a[i] * b[4 * j] + a[i + 4] * b[4 * j + 1] + a[i + 8] * b[4 * j + 2] + a[i + 12] * b[4 * j + 3];
    return r;
}

// XForm.h:474-484: homogeneous transform of a float point, evaluated in double, rounded per coordinate
static inline point operator*(const xform &xf, const point &p) {
    const double v0 = (double)p[0], v1 = (double)p[1], v2 = (double)p[2];
    const double h = 1.0 / (xf[3] * v0 + xf[7] * v1 + xf[11] * v2 + xf[15]);
    return point((float)(h * (xf[0] * v0 + xf[4] * v1 + xf[8] * v2 + xf[12])),
                 (float)(h * (xf[1] * v0 + xf[5] * v1 + xf[9] * v2 + xf[13])),
                 (float)(h * (xf[2] * v0 + xf[6] * v1 + xf[10] * v2 + xf[14])));
}

// XForm.h:487-500: row by row, stream default formatting
static inline std::ostream &operator<<(std::ostream &os, const xform &m) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            os << m[i + 4 * j];
            if (j == 3) os << std::endl; else os << " ";
        }
    return os;
}
// XForm.h:501-523: three rows are mandatory, the fourth may be absent
static inline std::istream &operator>>(std::istream &is, xform &m) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) is >> m[i + 4 * j];
    if (!is.good()) { m = xform::identity(); return is; }
    for (int j = 0; j < 4; ++j) is >> m[3 + 4 * j];
    if (!is.good()) { m[3] = m[7] = m[11] = 0.0; m[15] = 1.0; is.clear(); return is; }
    return is;
}

// The subset of trimesh2's TriMesh the solver side touches.
struct TriMesh {
    struct Face {
        int v[3];
        Face() { v[0] = v[1] = v[2] = 0; }
        Face(int a, int b, int c) { v[0] = a; v[1] = b; v[2] = c; }
        int &operator[](int i) { return v[i]; }
        const int &operator[](int i) const { return v[i]; }
        int indexof(int x) const { return v[0] == x ? 0 : v[1] == x ? 1 : v[2] == x ? 2 : -1; }
    };
    std::vector<point> vertices;
    std::vector<Face> faces;
    std::vector<std::vector<int> > adjacentfaces; // faces around each vertex, ascending face id
    std::vector<Face> across_edge;                // face opposite corner j of face f, -1 on the boundary
    void need_faces() {}
    void need_normals() {} // rendering only
    void need_tstrips() { need_across_edge(); }
    void need_adjacentfaces() {
        if (!adjacentfaces.empty() || faces.empty()) return;
        adjacentfaces.resize(vertices.size());
        for (size_t f = 0; f < faces.size(); ++f) for (int j = 0; j < 3; ++j) adjacentfaces[faces[f][j]].push_back((int)f);
    }
    // TriMesh_connectivity.cc:92-134: the neighbour across edge (v1,v2) is the first face around v1
    // that also touches v2 and runs the edge the other way
    void need_across_edge() {
        if (!across_edge.empty()) return;
        need_adjacentfaces();
        if (adjacentfaces.empty()) return;
        const int nf = (int)faces.size();
        across_edge.assign(nf, Face(-1, -1, -1));
        for (int i = 0; i < nf; ++i)
            for (int j = 0; j < 3; ++j) {
                if (across_edge[i][j] != -1) continue;
                const int v1 = faces[i][(j + 1) % 3], v2 = faces[i][(j + 2) % 3];
                const std::vector<int> &a1 = adjacentfaces[v1], &a2 = adjacentfaces[v2];
                for (size_t k = 0; k < a1.size(); ++k) {
                    const int other = a1[k];
                    if (other == i) continue;
                    if (std::find(a2.begin(), a2.end(), other) == a2.end()) continue;
                    const int ind = (faces[other].indexof(v1) + 1) % 3;
                    if (faces[other][(ind + 1) % 3] != v2) continue;
                    across_edge[i][j] = other;
                    across_edge[other][ind] = i;
                    break;
                }
            }
    }
};

static inline void apply_xform(TriMesh *mesh, const xform &xf) {
    for (size_t i = 0; i < mesh->vertices.size(); ++i) mesh->vertices[i] = xf * mesh->vertices[i];
}

// TriMeshBuilder.h:116-171: (tess_x+1)(tess_y+1) grid nodes, x outer / y inner, in [-1,1]^2 at z=0,
// then one centre node per cell; four faces per cell (ll,lr,c) (lr,ur,c) (c,ur,ul) (ll,c,ul).
// Coordinates are float expressions.
static inline void make_sym_plane(TriMesh *mesh, int tess_x, int tess_y) {
    if (tess_x < 1) tess_x = 1;
    if (tess_y < 1) tess_y = 1;
    mesh->vertices.reserve((tess_x + 1) * (tess_y + 1) + tess_x * tess_y);
    for (int x = 0; x < tess_x + 1; ++x)
        for (int y = 0; y < tess_y + 1; ++y) {
            const float xp = -1.0f + 2.0f * x / tess_x, yp = -1.0f + 2.0f * y / tess_y;
            mesh->vertices.push_back(point(xp, yp, 0));
        }
    for (int x = 0; x < tess_x; ++x)
        for (int y = 0; y < tess_y; ++y) {
            float xp = -1.0f + 2.0f * x / tess_x, yp = -1.0f + 2.0f * y / tess_y;
            xp += 1.f / tess_x;
            yp += 1.f / tess_y;
            mesh->vertices.push_back(point(xp, yp, 0));
        }
    mesh->faces.reserve(tess_x * tess_y * 4);
    for (int x = 0; x < tess_x; ++x)
        for (int y = 0; y < tess_y; ++y) {
            const int ll = y + x * (tess_y + 1), lr = y + (x + 1) * (tess_y + 1), ul = ll + 1, ur = lr + 1;
            const int cent = (tess_x + 1) * (tess_y + 1) + x * tess_y + y;
            mesh->faces.push_back(TriMesh::Face(ll, lr, cent));
            mesh->faces.push_back(TriMesh::Face(lr, ur, cent));
            mesh->faces.push_back(TriMesh::Face(cent, ur, ul));
            mesh->faces.push_back(TriMesh::Face(ll, cent, ul));
        }
}


// ---- mclscene's primitive objects (DefaultBuilders.hpp:83-256 call these): sphere, box / cube, beam, cylinder, torus --------------
// Restated from trimesh2's TriMeshBuilder.h (make_cube :220-340, make_sphere_polar :343-382, make_beam :385-422, make_ccyl :426-494,
// make_cyl :497-522, make_torus :525-556) and libsrc/remove.cc (remove_faces, remove_unused_vertices): the vertex ORDER and every float
// operation matter -- ForceBuilder numbers the solver's nodes by vertex index and lumps masses from these coordinates -- so the loops
// keep trimesh2's order and its arithmetic: angles are `constant_f * int / int` in float; the unqualified cos / sin of TriMeshBuilder.h
// resolve to the C library's DOUBLE functions (no `using namespace std` at that scope), so a float angle is promoted, `r * cos(ph)` is a
// double product, and the value is rounded to float only when it becomes a vertex coordinate -- except the torus tube, which calls cosf /
// sinf by name.  Settled against fixtures dumped from the reference's own loader (tests/golden/scene_shapes.npz): the float overloads differ
// from this in the last bit on 29 of 1227 coordinates of that scene.
namespace detail {
static const float kPif = 3.1415927f, kTwoPif = 6.2831855f;      // TriMesh.h:16-17 M_PIf, Color.h:25-26 M_TWOPIf
inline void quad(TriMesh *m, int ll, int lr, int ul, int ur) { m->faces.push_back(TriMesh::Face(ll, lr, ur)); m->faces.push_back(TriMesh::Face(ll, ur, ul)); }
inline int sq(int v) { return v * v; }
inline double dcos(float a) { return ::cos((double)a); }
inline double dsin(float a) { return ::sin((double)a); }
} // namespace detail

static inline void make_sphere_polar(TriMesh *mesh, int tess_ph, int tess_th) {
    using namespace detail;
    tess_th = std::max(tess_th, 3); tess_ph = std::max(tess_ph, 3);
    mesh->vertices.push_back(point(0, 0, -1));
    for (int j = 1; j < tess_th; ++j) {                       // rings from the south pole up
        const float th = kPif * j / tess_th;
        const float z = (float)-dcos(th), r = (float)dsin(th);
        for (int i = 0; i < tess_ph; ++i) { const float ph = kTwoPif * i / tess_ph; mesh->vertices.push_back(point((float)(r * dcos(ph)), (float)(r * dsin(ph)), z)); }
    }
    mesh->vertices.push_back(point(0, 0, 1));
    for (int i = 0; i < tess_ph; ++i) mesh->faces.push_back(TriMesh::Face(0, ((i + 1) % tess_ph) + 1, i + 1));
    for (int j = 0; j < tess_th - 2; ++j) {
        const int base = 1 + j * tess_ph;
        for (int i = 0; i < tess_ph; ++i) { const int i1 = (i + 1) % tess_ph; quad(mesh, base + i, base + i1, base + tess_ph + i, base + tess_ph + i1); }
    }
    const int base = 1 + (tess_th - 2) * tess_ph;
    for (int i = 0; i < tess_ph; ++i) mesh->faces.push_back(TriMesh::Face(base + i, base + ((i + 1) % tess_ph), base + tess_ph));
}

// [-1, 1]^3: the z = -1 face ((tess+1)^2 nodes, x and y running DOWN from +1), tess - 1 rings of 4 tess nodes around the sides (-y, +x, +y, -x side
// in turn), the z = +1 face (x and y running up); faces: bottom, the four sides ring by ring, top
static inline void make_cube(TriMesh *mesh, int tess) {
    using namespace detail;
    tess = std::max(tess, 1);
    const int t1 = tess + 1;
    for (int j = 0; j < t1; ++j) { const float y = 1.0f - 2.0f * j / tess; for (int i = 0; i < t1; ++i) { const float x = 1.0f - 2.0f * i / tess; mesh->vertices.push_back(point(x, y, -1)); } }
    for (int j = 1; j < tess; ++j) {
        const float z = -1.0f + 2.0f * j / tess;
        for (int i = 0; i < tess; ++i) { const float x = -1.0f + 2.0f * i / tess; mesh->vertices.push_back(point(x, -1, z)); }
        for (int i = 0; i < tess; ++i) { const float y = -1.0f + 2.0f * i / tess; mesh->vertices.push_back(point(1, y, z)); }
        for (int i = 0; i < tess; ++i) { const float x = 1.0f - 2.0f * i / tess; mesh->vertices.push_back(point(x, 1, z)); }
        for (int i = 0; i < tess; ++i) { const float y = 1.0f - 2.0f * i / tess; mesh->vertices.push_back(point(-1, y, z)); }
    }
    for (int j = 0; j < t1; ++j) { const float y = -1.0f + 2.0f * j / tess; for (int i = 0; i < t1; ++i) { const float x = -1.0f + 2.0f * i / tess; mesh->vertices.push_back(point(x, y, 1)); } }
    for (int j = 0; j < tess; ++j) for (int i = 0; i < tess; ++i) { const int ind = i + j * t1; quad(mesh, ind, ind + t1, ind + 1, ind + t1 + 1); }
    const int top = sq(t1) + 4 * tess * (tess - 1);
    for (int j = 0; j < tess; ++j) {
        int next = sq(t1) + 4 * tess * (j - 1);
        for (int side = 0; side < 4; ++side)
            for (int i = 0; i < tess; ++i) {
                int ll = next++, lr = ll + 1, ul = ll + 4 * tess, ur = ul + 1;
                if (j == 0) {                                  // the lower edge lies on the bottom face
                    if (side == 0) { ll = sq(t1) - 1 - i; lr = ll - 1; }
                    else if (side == 1) { ll = tess * t1 - i * t1; lr = ll - t1; }
                    else if (side == 2) { ll = i; lr = i + 1; }
                    else { ll = tess + i * t1; lr = ll + t1; }
                }
                if (j == tess - 1 && side > 0) {               // the upper edge lies on the top face (side 0's does too, but trimesh2 leaves that row to the ring formula)
                    if (side == 1) { ul = top + tess + i * t1; ur = ul + t1; }
                    else if (side == 2) { ul = top + sq(t1) - 1 - i; ur = ul - 1; }
                    else { ul = top + tess * t1 - i * t1; ur = ul - t1; }
                }
                if (side == 3 && i == tess - 1) { if (j != 0) lr -= 4 * tess; if (j != tess - 1) ur -= 4 * tess; }      // the ring closes
                quad(mesh, ll, lr, ul, ur);
            }
    }
    for (int j = 0; j < tess; ++j) for (int i = 0; i < tess; ++i) { const int ind = top + i + j * t1; quad(mesh, ind, ind + 1, ind + t1, ind + t1 + 1); }
}

// remove.cc: faces / vertices compacted in place, order kept
static inline void remove_faces(TriMesh *mesh, const std::vector<bool> &gone) {
    size_t next = 0;
    for (size_t f = 0; f < mesh->faces.size(); ++f) if (!gone[f]) mesh->faces[next++] = mesh->faces[f];
    mesh->faces.resize(next);
    mesh->adjacentfaces.clear(); mesh->across_edge.clear();
}
static inline void remove_unused_vertices(TriMesh *mesh) {
    const size_t nv = mesh->vertices.size();
    std::vector<int> remap(nv, -1);
    for (size_t f = 0; f < mesh->faces.size(); ++f) for (int j = 0; j < 3; ++j) remap[mesh->faces[f][j]] = 0;
    int next = 0;
    for (size_t v = 0; v < nv; ++v) if (remap[v] == 0) { remap[v] = next; mesh->vertices[next++] = mesh->vertices[v]; }
    mesh->vertices.resize(next);
    for (size_t f = 0; f < mesh->faces.size(); ++f) for (int j = 0; j < 3; ++j) mesh->faces[f][j] = remap[mesh->faces[f][j]];
    mesh->adjacentfaces.clear(); mesh->across_edge.clear();
}
// Vec.h:854-857 with operator% (:519-524): half the cross product of the two edges from corner 0, in float
static inline vec trinorm(const point &v0, const point &v1, const point &v2) {
    const vec a = v1 - v0, b = v2 - v0;
    return vec(0.5f * (a[1] * b[2] - a[2] * b[1]), 0.5f * (a[2] * b[0] - a[0] * b[2]), 0.5f * (a[0] * b[1] - a[1] * b[0]));
}

// `chunks` cubes in a row along +x (2 apart), the faces between neighbours dropped (a face whose normal points along -x / +x), unused vertices
// dropped per cube, the cubes appended one after the other WITHOUT merging the coincident rim nodes (mclscene's "box" is a beam of one chunk)
static inline void make_beam(TriMesh *mesh, int tess, int chunks) {
    for (int b = 0; b < chunks; ++b) {
        TriMesh box;
        make_cube(&box, tess);
        apply_xform(&box, xform::trans(b * 2.f, 0, 0));
        std::vector<bool> gone(box.faces.size(), false);
        for (size_t f = 0; f < box.faces.size(); ++f) {
            const vec n = trinorm(box.vertices[box.faces[f][0]], box.vertices[box.faces[f][1]], box.vertices[box.faces[f][2]]);
            if (b > 0 && n[0] * -1.f + n[1] * 0.f + n[2] * 0.f > 0.f) gone[f] = true;
            if (b < chunks - 1 && n[0] * 1.f + n[1] * 0.f + n[2] * 0.f > 0.f) gone[f] = true;
        }
        remove_faces(&box, gone);
        remove_unused_vertices(&box);
        const int off = (int)mesh->vertices.size();
        mesh->vertices.insert(mesh->vertices.end(), box.vertices.begin(), box.vertices.end());
        for (size_t f = 0; f < box.faces.size(); ++f) mesh->faces.push_back(TriMesh::Face(box.faces[f][0] + off, box.faces[f][1] + off, box.faces[f][2] + off));
    }
}

// capped cylinder about z in [-1, 1]: bottom centre, tess_h bottom rings (radius growing), tess_h - 1 side rings, tess_h top rings (radius shrinking), top centre
static inline void make_ccyl(TriMesh *mesh, int tess_th, int tess_h, float r = 1.0f) {
    using namespace detail;
    tess_th = std::max(tess_th, 3); tess_h = std::max(tess_h, 1);
    mesh->vertices.push_back(point(0, 0, -1));
    for (int j = 1; j <= tess_h; ++j) {
        const float rr = r * j / tess_h;
        for (int i = 0; i < tess_th; ++i) { const float th = kTwoPif * i / tess_th; mesh->vertices.push_back(point((float)(rr * dcos(th)), (float)(rr * dsin(th)), -1)); }
    }
    const int side_start = (int)mesh->vertices.size();
    for (int j = 1; j < tess_h; ++j) {
        const float z = -1.0f + 2.0f * j / tess_h;
        for (int i = 0; i < tess_th; ++i) { const float th = kTwoPif * i / tess_th; mesh->vertices.push_back(point((float)(r * dcos(th)), (float)(r * dsin(th)), z)); }
    }
    const int top_start = (int)mesh->vertices.size();
    for (int j = tess_h; j > 0; --j) {
        const float rr = r * j / tess_h;
        for (int i = 0; i < tess_th; ++i) { const float th = kTwoPif * i / tess_th; mesh->vertices.push_back(point((float)(rr * dcos(th)), (float)(rr * dsin(th)), 1)); }
    }
    mesh->vertices.push_back(point(0, 0, 1));
    for (int i = 0; i < tess_th; ++i) mesh->faces.push_back(TriMesh::Face(0, ((i + 1) % tess_th) + 1, i + 1));
    for (int j = 1; j < tess_h; ++j) {
        const int base = 1 + (j - 1) * tess_th;
        for (int i = 0; i < tess_th; ++i) { const int i1 = (i + 1) % tess_th; quad(mesh, base + tess_th + i1, base + tess_th + i, base + i1, base + i); }
    }
    for (int j = 0; j < tess_h; ++j) {
        const int base = side_start - tess_th + j * tess_th;
        for (int i = 0; i < tess_th; ++i) { const int i1 = (i + 1) % tess_th; quad(mesh, base + i, base + i1, base + tess_th + i, base + tess_th + i1); }
    }
    for (int j = 0; j < tess_h - 1; ++j) {
        const int base = top_start + j * tess_th;
        for (int i = 0; i < tess_th; ++i) { const int i1 = (i + 1) % tess_th; quad(mesh, base + tess_th + i1, base + tess_th + i, base + i1, base + i); }
    }
    const int base = top_start + (tess_h - 1) * tess_th;
    for (int i = 0; i < tess_th; ++i) mesh->faces.push_back(TriMesh::Face(base + i, base + ((i + 1) % tess_th), base + tess_th));
}

// open cylinder: tess_h + 1 rings of tess_th nodes
static inline void make_cyl(TriMesh *mesh, int tess_th, int tess_h, float r) {
    using namespace detail;
    tess_th = std::max(tess_th, 3); tess_h = std::max(tess_h, 1);
    for (int j = 0; j <= tess_h; ++j) {
        const float z = -1.0f + 2.0f * j / tess_h;
        for (int i = 0; i < tess_th; ++i) { const float th = kTwoPif * i / tess_th; mesh->vertices.push_back(point((float)(r * dcos(th)), (float)(r * dsin(th)), z)); }
    }
    for (int j = 0; j < tess_h; ++j) {
        const int base = j * tess_th;
        for (int i = 0; i < tess_th; ++i) { const int i1 = (i + 1) % tess_th; quad(mesh, base + i, base + i1, base + tess_th + i, base + tess_th + i1); }
    }
}

// a cylinder's connectivity with its last ring identified with the first, the nodes moved onto the torus of tube radius inner_rad about the unit circle
// (outer_rad only sizes the discarded cylinder: "doesn't do anything" as DefaultBuilders.hpp:237 remarks)
static inline void make_torus(TriMesh *mesh, int tess_th, int tess_ph, float inner_rad, float outer_rad) {
    using namespace detail;
    tess_th = std::max(tess_th, 3); tess_ph = std::max(tess_ph, 3);
    make_cyl(mesh, tess_ph, tess_th, outer_rad);
    mesh->vertices.resize(mesh->vertices.size() - tess_ph);
    const int nv = (int)mesh->vertices.size();
    for (size_t f = 0; f < mesh->faces.size(); ++f) for (int j = 0; j < 3; ++j) mesh->faces[f][j] %= nv;
    const float r = inner_rad;
    for (int j = 0; j < tess_th; ++j) {
        const float th = kTwoPif * j / tess_th;
        const vec circlepos((float)dcos(th), (float)dsin(th), 0);
        for (int i = 0; i < tess_ph; ++i) {
            const float ph = kTwoPif * i / tess_ph;
            const float cr = cosf(ph) * r, sr = sinf(ph) * r;
            const vec a(cr * circlepos[0], cr * circlepos[1], cr * circlepos[2]), b(sr * 0.f, sr * 0.f, sr * -1.f);
            mesh->vertices[i + j * tess_ph] = (circlepos + a) + b;
        }
    }
}


// ---- "trimesh" objects from a Wavefront OBJ file (trimesh2 libsrc/TriMesh_io.cc: read_helper :232-337, read_obj :736-788, tess :1374-1407,
// skip_comments :1352-1370); PLY below.  Same parsing: the file type is sniffed from
// the first byte ('#', or one of v u f g s o); after a leading '#' ONE word is consumed and the rest of that line is read as an ordinary line;
// `v x y z` through sscanf("%f %f %f") into float coordinates; `f` / `t` lines take the integer at the start of every whitespace-separated
// token (so `f 1/2/3 4/5/6 ...` reads the position indices), 1-based or negative (relative to the vertices read so far); quads are cut along
// their shorter diagonal (float distances, ties: the 1-3 diagonal), larger polygons as a fan from their first corner; lines are read in pieces of
// at most 1023 bytes like fgets(buf, 1024).  Indices outside the vertex list are an error here (trimesh2's check_ind_range warns and guesses).
// an n-gon -> triangles (TriMesh_io.cc tess :1374-1407): quads along their shorter diagonal (float distances, ties: the 1-3 diagonal), larger polygons as a fan
static inline void tess_polygon(const std::vector<point> &verts, const std::vector<int> &c, std::vector<TriMesh::Face> &tris) {
    const size_t nc = c.size();
    if (nc < 3) return;
    if (nc == 3) { tris.push_back(TriMesh::Face(c[0], c[1], c[2])); return; }
    if (nc == 4) {
        const vec d02 = verts[c[0]] - verts[c[2]], d13 = verts[c[1]] - verts[c[3]];
        const int i = (len2(d02) < len2(d13)) ? 0 : 1;
        tris.push_back(TriMesh::Face(c[i], c[(i + 1) % 4], c[(i + 2) % 4]));
        tris.push_back(TriMesh::Face(c[i], c[(i + 2) % 4], c[(i + 3) % 4]));
        return;
    }
    for (size_t k = 2; k < nc; ++k) tris.push_back(TriMesh::Face(c[0], c[k - 1], c[k]));
}
// skip_comments (:1352-1370): blank space and '#' ... end of line
static inline void skip_comments(FILE *f) {
    bool in_comment = false;
    for (;;) {
        const int c = std::fgetc(f);
        if (c == EOF) return;
        if (in_comment) { if (c == '\n') in_comment = false; }
        else if (c == '#') in_comment = true;
        else if (!std::isspace(c)) { std::ungetc(c, f); return; }
    }
}

static inline bool read_obj_body(FILE *f, const char *filename, TriMesh *mesh, std::string *why) {
    bool ok = true;
    std::vector<int> corners;
    char buf[1024];
    while (ok) {
        skip_comments(f);
        { const int c = std::fgetc(f); if (c == EOF) break; std::ungetc(c, f); }
        if (std::feof(f)) break;
        if (!std::fgets(buf, 1024, f)) { ok = false; break; }
        auto is = [&](const char *t) { return strncasecmp(buf, t, std::strlen(t)) == 0; };
        if (is("v ") || is("v\t")) {
            float x, y, z;
            if (std::sscanf(buf + 1, "%f %f %f", &x, &y, &z) != 3) { ok = false; break; }
            mesh->vertices.push_back(point(x, y, z));
        } else if (is("f ") || is("f\t") || is("t ") || is("t\t")) {
            corners.clear();
            char *q = buf;
            for (;;) {
                while (*q && *q != '\n' && !std::isspace((unsigned char)*q)) ++q;
                while (*q && std::isspace((unsigned char)*q)) ++q;
                int idx;
                if (std::sscanf(q, " %d", &idx) != 1) break;
                corners.push_back(idx < 0 ? idx + (int)mesh->vertices.size() : idx - 1);
            }
            for (size_t k = 0; k < corners.size(); ++k) if (corners[k] < 0 || corners[k] >= (int)mesh->vertices.size()) { if (why) *why = std::string(filename) + ": face index outside the vertices read so far"; ok = false; }
            if (!ok) break;
            tess_polygon(mesh->vertices, corners, mesh->faces);
        }
    }
    if (!ok && why && why->empty()) *why = std::string("error reading ") + filename;
    return ok;
}

// ---- PLY (TriMesh_io.cc read_ply :342-556 with read_verts_asc / _bin, read_faces_asc / _bin, ply_type_len, check_need_swap): ascii and binary of
// either byte order; the vertex element's position is three consecutive floats starting at `property float x`; whatever other scalar properties and
// whatever elements precede the vertices or sit between vertices and faces are skipped by their declared sizes (words in ascii files); faces are
// `property list <count type> <index type> vertex_ind...` lists -- the count read as 1 or 4 bytes, the indices ALWAYS as 4-byte integers, like
// trimesh2 does -- plus other scalar properties; polygons are cut like OBJ faces.  Triangle strips and range grids are not carried.
namespace detail {
inline int ply_type_len(const char *t, bool binary) {
    auto is = [&](const char *w) { return strncasecmp(t, w, std::strlen(w)) == 0; };
    if (is("char") || is("uchar") || is("int8") || is("uint8")) return 1;
    if (is("short") || is("ushort") || is("int16") || is("uint16")) return binary ? 2 : 1;
    if (is("int") || is("uint") || is("float") || is("int32") || is("uint32") || is("float32")) return binary ? 4 : 1;
    if (is("double") || is("float64")) return binary ? 8 : 1;
    return 0;
}
inline void swap32(void *p) { unsigned char *c = (unsigned char *)p; std::swap(c[0], c[3]); std::swap(c[1], c[2]); }
} // namespace detail

static inline bool read_ply_body(FILE *f, const char *filename, TriMesh *mesh, std::string *why) {
    using namespace detail;
    char buf[1024];
    auto fail = [&](const char *msg) { if (why) *why = std::string(filename) + ": " + msg; return false; };
    auto line = [&]() { return std::fgets(buf, 1024, f) != 0; };
    auto is = [&](const char *t) { return strncasecmp(buf, t, std::strlen(t)) == 0; };
    const int one = 1; const bool little = *(const unsigned char *)&one != 0;
    if (!line()) return fail("truncated header");
    while (buf[0] && std::isspace((unsigned char)buf[0])) if (!line()) return fail("truncated header");
    bool binary = false, need_swap = false;
    if (is("format binary_big_endian 1.0")) { binary = true; need_swap = little; }
    else if (is("format binary_little_endian 1.0")) { binary = true; need_swap = !little; }
    else if (!is("format ascii 1.0")) return fail("unknown ply format or version");
    if (!line()) return fail("truncated header");
    while (is("obj_info") || is("comment")) if (!line()) return fail("truncated header");
    // elements ahead of the vertices / between vertices and faces: sized and skipped
    auto skip_elements = [&](int &skip, std::initializer_list<const char *> stops) -> bool {
        for (;;) {
            for (const char *st : stops) if (is(st)) return true;
            char name[1024]; int nelem = 0, elem_len = 0;
            std::sscanf(buf, "element %1023s %d", name, &nelem);
            if (!line()) return false;
            while (is("property")) { const int tl = ply_type_len(buf + 9, binary); if (!tl) return false; elem_len += tl; if (!line()) return false; }
            skip += nelem * elem_len;
        }
    };
    int skip1 = 0, skip2 = 0, nverts = 0, nfaces = 0;
    if (!skip_elements(skip1, {"end_header", "element vertex"})) return fail("unsupported property ahead of the vertices");
    if (std::sscanf(buf, "element vertex %d\n", &nverts) != 1) return fail("expected \"element vertex\"");
    int vert_len = 0, vert_pos = -1, vert_norm = -1, vert_color = -1, vert_conf = -1; bool float_color = false;
    if (!line()) return fail("truncated header");
    while (is("property")) {
        if (is("property float x") || is("property float32 x")) vert_pos = vert_len;
        if (is("property float nx") || is("property float32 nx")) vert_norm = vert_len;
        if (is("property uchar diffuse_red") || is("property uint8 diffuse_red") || is("property uchar red") || is("property uint8 red")) vert_color = vert_len;
        if (is("property float diffuse_red") || is("property float32 diffuse_red") || is("property float red") || is("property float32 red")) { vert_color = vert_len; float_color = true; }
        if (is("property float confidence") || is("property float32 confidence")) vert_conf = vert_len;
        const int tl = ply_type_len(buf + 9, binary);
        if (!tl) return fail("unsupported vertex property");
        vert_len += tl;
        if (!line()) return fail("truncated header");
    }
    if (!skip_elements(skip2, {"end_header", "element face", "element tristrips", "element range_grid"})) return fail("unsupported property between vertices and faces");
    int face_len = 0, face_count = -1, face_idx = -1;
    if (is("element face")) {
        if (std::sscanf(buf, "element face %d\n", &nfaces) != 1) return fail("bad face element");
        if (!line()) return fail("truncated header");
        while (is("property")) {
            char ct[256], it[256];
            if (std::sscanf(buf, "property list %255s %255s vertex_ind", ct, it) == 2) {
                const int cl = ply_type_len(ct, binary), il = ply_type_len(it, binary);
                if (cl && il) { face_count = face_len; face_idx = face_len + cl; face_len += cl; }
            } else { const int tl = ply_type_len(buf + 9, binary); if (!tl) return fail("unsupported face property"); face_len += tl; }
            if (!line()) return fail("truncated header");
        }
    } else if (is("element tristrips") || is("element range_grid")) return fail("triangle strips / range grids are not carried by this loader");
    while (!is("end_header")) if (!line()) return fail("no end_header");
    // ---- data ----
    auto skip_data = [&](int n) { if (binary) std::fseek(f, n, SEEK_CUR); else for (int i = 0; i < n; ++i) if (std::fscanf(f, "%1023s", buf) != 1) break; };
    if (skip1) skip_data(skip1);
    if (nverts <= 0 || vert_pos < 0 || vert_len < (binary ? 12 : 3)) return fail("no float x y z vertices");
    const size_t v0 = mesh->vertices.size();
    mesh->vertices.resize(v0 + nverts);
    if (binary) {
        std::vector<unsigned char> rec(vert_len);
        for (int i = 0; i < nverts; ++i) {
            if (std::fread(rec.data(), vert_len, 1, f) != 1) return fail("truncated vertex data");
            float q[3]; std::memcpy(q, &rec[vert_pos], 12);
            if (i == 0) {      // check_need_swap: a first vertex that only makes sense in the other byte order flips the declared one
                auto sane = [](const float *w) { return w[0] > -1.0e10f && w[0] < 1.0e10f && w[1] > -1.0e10f && w[1] < 1.0e10f && w[2] > -1.0e10f && w[2] < 1.0e10f; };
                float t[3] = {q[0], q[1], q[2]};
                if (need_swap) for (int k = 0; k < 3; ++k) swap32(&t[k]);
                if (!sane(t)) { for (int k = 0; k < 3; ++k) swap32(&t[k]); if (sane(t)) need_swap = !need_swap; }
            }
            if (need_swap) for (int k = 0; k < 3; ++k) swap32(&q[k]);
            mesh->vertices[v0 + i] = point(q[0], q[1], q[2]);
        }
    } else {
        skip_comments(f);
        for (int i = 0; i < nverts; ++i)
            for (int j = 0; j < vert_len; ++j) {
                float a, b, c; int ia, ib, ic;
                if (j == vert_pos) { if (std::fscanf(f, "%f %f %f", &a, &b, &c) != 3) return fail("bad vertex"); mesh->vertices[v0 + i] = point(a, b, c); j += 2; }
                else if (j == vert_norm) { if (std::fscanf(f, "%f %f %f", &a, &b, &c) != 3) return fail("bad normal"); j += 2; }
                else if (j == vert_color && float_color) { if (std::fscanf(f, "%f %f %f", &a, &b, &c) != 3) return fail("bad colour"); j += 2; }
                else if (j == vert_color) { if (std::fscanf(f, "%d %d %d", &ia, &ib, &ic) != 3) return fail("bad colour"); j += 2; }
                else if (j == vert_conf) { if (std::fscanf(f, "%f", &a) != 1) return fail("bad confidence"); }
                else if (std::fscanf(f, " %1023s", buf) != 1) return fail("truncated vertex data");
            }
    }
    if (skip2) skip_data(skip2);
    if (nfaces > 0) {
        if (face_idx < 0) return fail("faces without a vertex index list");
        std::vector<int> corners;
        if (binary) {
            const int face_skip = face_len - face_idx;
            std::vector<unsigned char> rec(std::max(std::max(face_idx, face_skip), 4));
            for (int i = 0; i < nfaces; ++i) {
                if (face_idx > 0 && std::fread(rec.data(), face_idx, 1, f) != 1) return fail("truncated face data");
                unsigned n = 3;
                if (face_count >= 0) { if (face_idx - face_count == 4) { std::memcpy(&n, &rec[face_count], 4); if (need_swap) swap32(&n); } else n = rec[face_count]; }
                if (n > 1000000u) return fail("implausible polygon size");
                corners.resize(n);
                if (n && std::fread(corners.data(), 4 * (size_t)n, 1, f) != 1) return fail("truncated face data");
                if (need_swap) for (size_t k = 0; k < corners.size(); ++k) swap32(&corners[k]);
                                for (size_t k = 0; k < corners.size(); ++k) if (corners[k] < 0 || corners[k] > (int)mesh->vertices.size()) return fail("face index outside the vertex list");
                if (corners.size() == 4) for (size_t k = 0; k < 4; ++k) if (corners[k] >= (int)mesh->vertices.size()) return fail("face index outside the vertex list");
                tess_polygon(mesh->vertices, corners, mesh->faces);
                if (face_skip > 0 && std::fread(rec.data(), face_skip, 1, f) != 1) return fail("truncated face data");
            }
        } else {
            skip_comments(f);
            for (int i = 0; i < nfaces; ++i) {
                corners.clear();
                int count = 3;
                for (int j = 0; j < face_len + count; ++j) {
                    if (j >= face_idx && j < face_idx + count) { int v = 0; if (std::fscanf(f, " %d", &v) != 1) return fail("bad face index"); corners.push_back(v); }
                    else if (j == face_count) { if (std::fscanf(f, " %d", &count) != 1) return fail("bad face count"); }
                    else if (std::fscanf(f, " %1023s", buf) != 1) return fail("truncated face data");
                }
                if (corners.size() == 4) for (size_t k = 0; k < 4; ++k) if (corners[k] < 0 || corners[k] >= (int)mesh->vertices.size()) return fail("face index outside the vertex list");
                tess_polygon(mesh->vertices, corners, mesh->faces);
            }
        }
    }
    return true;
}

// OFF (read_off :792-806): counts line, `x y z` per vertex, `n i0 .. i(n-1)` per face, the rest of a face's line (colours) ignored
static inline bool read_off_body(FILE *f, const char *filename, TriMesh *mesh, std::string *why) {
    auto fail = [&](const char *msg) { if (why) *why = std::string(filename) + ": " + msg; return false; };
    char buf[1024];
    skip_comments(f);
    if (!std::fgets(buf, 1024, f)) return fail("truncated header");
    int nverts = 0, nfaces = 0, unused = 0;
    if (std::sscanf(buf, "%d %d %d", &nverts, &nfaces, &unused) < 2 || nverts <= 0 || nfaces < 0) return fail("bad OFF counts");
    skip_comments(f);
    for (int i = 0; i < nverts; ++i) { float a, b, c; if (std::fscanf(f, "%f %f %f", &a, &b, &c) != 3) return fail("bad vertex"); mesh->vertices.push_back(point(a, b, c)); }
    if (nfaces) skip_comments(f);
    std::vector<int> corners;
    for (int i = 0; i < nfaces; ++i) {
        int count = 3;
        if (std::fscanf(f, " %d", &count) != 1 || count < 0 || count > 1000000) return fail("bad face count");
        corners.clear();
        for (int j = 0; j < count; ++j) { int v = 0; if (std::fscanf(f, " %d", &v) != 1) return fail("bad face index"); corners.push_back(v); }
        if (corners.size() == 4) for (size_t k = 0; k < 4; ++k) if (corners[k] < 0 || corners[k] >= (int)mesh->vertices.size()) return fail("face index outside the vertex list");
        tess_polygon(mesh->vertices, corners, mesh->faces);
        for (;;) { const int c = std::fgetc(f); if (c == EOF || c == '\n') break; }
    }
    return true;
}

// TriMesh::read_helper (:232-337): the file type from its first byte(s); afterwards check_ind_range (:1316-1347) -- indices that run 1..N (or k..k+N-1) are shifted to 0..N-1
static inline bool read_mesh_file(const char *filename, TriMesh *mesh, std::string *why) {
    FILE *f = std::fopen(filename, "rb");
    if (!f) { if (why) *why = std::string("cannot open ") + filename; return false; }
    bool ok = false;
    const int c = std::fgetc(f);
    if (c == 'p') { char b[4]; if (std::fgets(b, 4, f) && std::strncmp(b, "ly", 2) == 0) ok = read_ply_body(f, filename, mesh, why); else if (why) *why = std::string(filename) + ": unknown file type"; }
    else if (c == '#') { char word[1025]; if (std::fscanf(f, "%1024s", word) != 1) word[0] = 0; ok = read_obj_body(f, filename, mesh, why); }
    else if (c == 'v' || c == 'u' || c == 'f' || c == 'g' || c == 's' || c == 'o') { std::ungetc(c, f); ok = read_obj_body(f, filename, mesh, why); }
    else if (c == 'O') { char b[3]; if (std::fgets(b, 3, f) && std::strncmp(b, "FF", 2) == 0) ok = read_off_body(f, filename, mesh, why); else if (why) *why = std::string(filename) + ": unknown file type"; }
    else if (why) *why = std::string(filename) + ": not a Wavefront OBJ, PLY or OFF file (the other formats of trimesh2's reader -- 3ds, stl, sm, vvd, ray -- are not carried)";
    std::fclose(f);
    if (ok && mesh->vertices.empty()) { ok = false; if (why) *why = std::string(filename) + ": no vertices"; }
    if (!ok) { if (why && why->empty()) *why = std::string("error reading ") + filename; return false; }
    if (!mesh->faces.empty()) {
        int lo = mesh->faces[0][0], hi = lo;
        for (size_t i = 0; i < mesh->faces.size(); ++i) for (int j = 0; j < 3; ++j) { lo = std::min(lo, mesh->faces[i][j]); hi = std::max(hi, mesh->faces[i][j]); }
        const int nv = (int)mesh->vertices.size();
        if (!(lo == 0 && hi == nv - 1)) {
            if (hi - lo == nv - 1) { for (size_t i = 0; i < mesh->faces.size(); ++i) for (int j = 0; j < 3; ++j) mesh->faces[i][j] -= lo; }
            else if (lo < 0 || hi >= nv) { if (why) *why = std::string(filename) + ": face indices outside the vertex list"; return false; }      // (trimesh2 goes on with them: undefined behaviour there)
        }
    }
    return true;
}

} // namespace trimesh

namespace mcl {

namespace parse {
static inline std::string to_lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }
static inline std::string fileDir(std::string fname) {
    const size_t pos = fname.find_last_of('/');
    return (std::string::npos == pos) ? "" : fname.substr(0, pos) + '/';
}
} // namespace parse

// ---- the XML subset the scene files use: nested elements with attributes, comments, a declaration.
// ---- Like the reference's pugixml use, several top-level elements are allowed and text is ignored.
namespace xml {
struct Node {
    std::string name;
    std::vector<std::pair<std::string, std::string> > attrs;
    std::vector<Node> children;
    // value of attribute `key`, "" if absent (pugi::xml_attribute::as_string/value of a null attribute)
    std::string attribute(const std::string &key) const {
        for (size_t i = 0; i < attrs.size(); ++i) if (attrs[i].first == key) return attrs[i].second;
        return "";
    }
};

class Reader {
public:
    explicit Reader(const std::string &text) : s(text), p(0) {}
    // children of a synthetic document node
    bool parse(Node &doc) {
        while (true) {
            skip_misc();
            if (p >= s.size()) return true;
            if (s[p] != '<') { ++p; continue; }
            Node n;
            if (!element(n)) return false;
            doc.children.push_back(n);
        }
    }
private:
    const std::string &s;
    size_t p;
    bool starts(const char *t) const { return s.compare(p, std::char_traits<char>::length(t), t) == 0; }
    void skip_ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) ++p; }
    // whitespace, text, comments, <? ?> and <! > outside tags
    void skip_misc() {
        while (p < s.size()) {
            if (starts("<!--")) { const size_t e = s.find("-->", p + 4); p = (e == std::string::npos) ? s.size() : e + 3; }
            else if (starts("<?")) { const size_t e = s.find("?>", p + 2); p = (e == std::string::npos) ? s.size() : e + 2; }
            else if (starts("<!")) { const size_t e = s.find('>', p + 2); p = (e == std::string::npos) ? s.size() : e + 1; }
            else if (s[p] == '<') return;
            else ++p;
        }
    }
    static bool name_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
    std::string name() { const size_t b = p; while (p < s.size() && name_char(s[p])) ++p; return s.substr(b, p - b); }
    static std::string unescape(const std::string &v) {
        if (v.find('&') == std::string::npos) return v;
        static const char *ent[5] = {"&amp;", "&lt;", "&gt;", "&quot;", "&apos;"};
        static const char rep[5] = {'&', '<', '>', '"', '\''};
        std::string o;
        for (size_t i = 0; i < v.size();) {
            bool hit = false;
            if (v[i] == '&')
                for (int e = 0; e < 5 && !hit; ++e) {
                    const size_t n = std::char_traits<char>::length(ent[e]);
                    if (v.compare(i, n, ent[e]) == 0) { o += rep[e]; i += n; hit = true; }
                }
            if (!hit) o += v[i++];
        }
        return o;
    }
    bool element(Node &n) {
        ++p; // '<'
        n.name = name();
        if (n.name.empty()) return false;
        while (true) {
            skip_ws();
            if (p >= s.size()) return false;
            if (starts("/>")) { p += 2; return true; }
            if (s[p] == '>') { ++p; break; }
            const std::string key = name();
            if (key.empty()) return false;
            skip_ws();
            if (p >= s.size() || s[p] != '=') return false;
            ++p; skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) return false;
            const char q = s[p++];
            const size_t e = s.find(q, p);
            if (e == std::string::npos) return false;
            n.attrs.push_back(std::make_pair(key, unescape(s.substr(p, e - p))));
            p = e + 1;
        }
        while (true) { // content
            skip_misc();
            if (p >= s.size()) return false;
            if (starts("</")) { const size_t e = s.find('>', p); if (e == std::string::npos) return false; p = e + 1; return true; }
            Node c;
            if (!element(c)) return false;
            n.children.push_back(c);
        }
    }
};

static inline bool load_file(const std::string &filename, Node &doc) {
    std::ifstream f(filename.c_str(), std::ios::in | std::ios::binary);
    if (!f) return false;
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    Reader r(text);
    return r.parse(doc);
}
// first top-level element whose lower-cased name is `lname`, or an empty node
static inline const Node *find_head(const Node &doc, const std::string &lname) {
    for (size_t i = 0; i < doc.children.size(); ++i) if (parse::to_lower(doc.children[i].name) == lname) return &doc.children[i];
    return 0;
}
} // namespace xml

// A parameter is (lower-case tag, value text); conversions go through a stringstream (Param.hpp:74-99,209-236)
class Param {
public:
    Param(std::string tag_, std::string value_) : tag(tag_), value(value_) {}
    double as_double() const { std::stringstream ss(value); double v; ss >> v; return v; }
    char as_char() const { std::stringstream ss(value); char v; ss >> v; return v; }
    std::string as_string() const { return value; }
    int as_int() const { std::stringstream ss(value); int v; ss >> v; return v; }
    long as_long() const { std::stringstream ss(value); long v; ss >> v; return v; }
    bool as_bool() const { std::stringstream ss(value); bool v; ss >> v; return v; }
    float as_float() const { std::stringstream ss(value); float v; ss >> v; return v; }
    trimesh::vec4 as_vec4() const { std::stringstream ss(value); trimesh::vec4 v; for (int i = 0; i < 4; ++i) ss >> v[i]; return v; }
    trimesh::vec as_vec3() const { std::stringstream ss(value); trimesh::vec v; for (int i = 0; i < 3; ++i) ss >> v[i]; return v; }
    trimesh::vec2 as_vec2() const { std::stringstream ss(value); trimesh::vec2 v; for (int i = 0; i < 2; ++i) ss >> v[i]; return v; }
    trimesh::xform as_xform() const { trimesh::xform x; std::stringstream ss(value); ss >> x; return x; }
    std::string tag;
    std::string value;
};

class Component {
public:
    Component(std::string tag_, std::string name_, std::string type_) : tag(tag_), name(name_), type(type_) {}
    std::string tag, name, type;
    // a missing parameter is appended with an empty value (Param.hpp:268-275)
    Param &get(std::string t) {
        for (size_t i = 0; i < params.size(); ++i) if (params[i].tag == t) return params[i];
        params.push_back(Param(t, ""));
        return params.back();
    }
    Param &operator[](std::string t) { return get(t); }
    bool exists(std::string t) const {
        for (size_t i = 0; i < params.size(); ++i) if (params[i].tag == t) return true;
        return false;
    }
    std::vector<Param> params;
};

// Param.hpp:122-166.  scale / translate / rotate values are replaced by the TEXT of the 4x4 matrix.
static inline void load_params(std::vector<Param> &params, const xml::Node &node) {
    for (size_t c = 0; c < node.children.size(); ++c) {
        const std::string tag = parse::to_lower(node.children[c].name);
        const std::string value = node.children[c].attribute("value");
        Param p(tag, value);
        if (tag == "scale" || tag == "translate" || tag == "rotate") {
            std::stringstream ss(value);
            trimesh::vec v; ss >> v[0] >> v[1] >> v[2];
            trimesh::xform xf;
            if (tag == "scale") xf = trimesh::xform::scale(v[0], v[1], v[2]);
            else if (tag == "translate") xf = trimesh::xform::trans(v[0], v[1], v[2]);
            else {
                const float to_rad = (float)(M_PI / 180.f); // degrees -> radians on float components
                for (int i = 0; i < 3; ++i) v[i] *= to_rad;
                trimesh::xform rot;
                rot = rot * trimesh::xform::rot(v[0], trimesh::vec(1.f, 0.f, 0.f));
                rot = rot * trimesh::xform::rot(v[1], trimesh::vec(0.f, 1.f, 0.f));
                rot = rot * trimesh::xform::rot(v[2], trimesh::vec(0.f, 0.f, 1.f));
                xf = rot;
            }
            std::stringstream out; out << xf;
            p.value = out.str();
        }
        params.push_back(p);
    }
}

class BaseObject {
public:
    virtual ~BaseObject() {}
    virtual std::string get_type() const = 0;
    virtual void update() {}
    virtual const std::shared_ptr<trimesh::TriMesh> get_TriMesh() { return std::shared_ptr<trimesh::TriMesh>(); }
    virtual void apply_xform(const trimesh::xform &) {}
    virtual std::string get_material() const { return ""; }
    virtual void set_material(std::string) {}
};

class TriangleMesh : public BaseObject {
public:
    TriangleMesh(std::shared_ptr<trimesh::TriMesh> tm, std::string mat = "") : tris(tm), vertices(tm->vertices), faces(tm->faces), material(mat) {}
    std::string get_type() const { return "trimesh"; }
    const std::shared_ptr<trimesh::TriMesh> get_TriMesh() { return tris; }
    void apply_xform(const trimesh::xform &xf) { trimesh::apply_xform(tris.get(), xf); }
    std::string get_material() const { return material; }
    void set_material(std::string mat) { material = mat; }
    std::shared_ptr<trimesh::TriMesh> tris;
    std::vector<trimesh::point> &vertices;
    std::vector<trimesh::TriMesh::Face> &faces;
private:
    std::string material;
};

// Scenery this loader does not build geometry for (mesh files read by trimesh2, point clouds): parameters only.
class StaticShape : public BaseObject {
public:
    StaticShape(std::string type_, std::string mat = "") : type(type_), material(mat) {}
    std::string get_type() const { return type; }
    std::string get_material() const { return material; }
    void set_material(std::string mat) { material = mat; }
private:
    std::string type, material;
};

// sorted vertex triple + the byte hash mclscene keys its face-count table with
// (include/MCL/VertexSort.hpp:55-75,103-112); the table's iteration order is the order of a tet mesh's surface faces
struct int3 {
    int3() {}
    int3(int a, int b, int c) {
        sorted_v[0] = a; sorted_v[1] = b; sorted_v[2] = c;
        if (sorted_v[0] > sorted_v[1]) std::swap(sorted_v[0], sorted_v[1]);
        if (sorted_v[0] > sorted_v[2]) std::swap(sorted_v[0], sorted_v[2]);
        if (sorted_v[1] > sorted_v[2]) std::swap(sorted_v[1], sorted_v[2]);
        orig_v[0] = a; orig_v[1] = b; orig_v[2] = c;
    }
    bool operator==(const int3 &o) const { return sorted_v[0] == o.sorted_v[0] && sorted_v[1] == o.sorted_v[1] && sorted_v[2] == o.sorted_v[2]; }
    int sorted_v[3], orig_v[3];
};
struct int3_hash {
    size_t operator()(const int3 &k) const {
        const unsigned char *in = reinterpret_cast<const unsigned char *>(k.sorted_v);
        unsigned int ret = 2654435761u;
        for (size_t i = 0; i < 3 * sizeof(int); ++i) ret = (ret * 2654435761u) ^ *in++;
        return ret;
    }
};

class TetMesh : public BaseObject {
    std::shared_ptr<trimesh::TriMesh> tris; // vertex / surface container
public:
    struct tet {
        tet() {}
        tet(int a, int b, int c, int d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
        int v[4];
    };
    std::vector<tet> tets;
    std::vector<trimesh::point> &vertices;        // ALL nodes, as float
    std::vector<trimesh::TriMesh::Face> &faces;   // boundary triangles
    TetMesh(std::string mat = "") : tris(new trimesh::TriMesh), vertices(tris->vertices), faces(tris->faces), material(mat) {}
    std::string get_type() const { return "tetmesh"; }
    const std::shared_ptr<trimesh::TriMesh> get_TriMesh() { return tris; }
    void apply_xform(const trimesh::xform &xf) { trimesh::apply_xform(tris.get(), xf); }
    std::string get_material() const { return material; }
    void set_material(std::string mat) { material = mat; }
    void need_normals(bool = true) {}

    // `filename` without extension; TetGen .node/.ele (0- or 1-based ids). ply input needs tetgen: not here.
    bool load(std::string filename) {
        vertices.clear(); tets.clear(); faces.clear();
        const size_t dot = filename.find_last_of('.');
        if (dot != std::string::npos && parse::to_lower(filename.substr(dot + 1)) == "ply") {
            std::cerr << "\n**TetMesh Error: tetrahedralising " << filename << " needs tetgen, which this headless loader does not carry" << std::endl;
            return false;
        }
        return load_node(filename) && load_ele(filename) && need_surface();
    }

private:
    std::string material;

    // TetMesh.cpp:133-178: "<count> ..." header line, then "<id> <x> <y> <z>" per line
    bool load_node(const std::string &filename) {
        const std::string path = filename + ".node";
        std::ifstream in(path.c_str());
        if (!in) { std::cerr << "\n**TetMesh Error: Could not load " << path << std::endl; return false; }
        std::string line;
        std::getline(in, line);
        int n_nodes = 0; { std::stringstream hs(line); hs >> n_nodes; }
        vertices.resize(n_nodes);
        std::vector<int> seen(n_nodes, 0);
        bool one_based = false;
        for (int i = 0; i < n_nodes; ++i) {
            std::getline(in, line);
            std::stringstream ls(line);
            double x, y, z; int id;
            ls >> id >> x >> y >> z;
            if (i == 0 && id == 1) one_based = true;
            if (one_based) id -= 1;
            if (id < 0 || id >= n_nodes) { std::cerr << "\n**TetMesh Error: Your indices are bad for file " << path << std::endl; return false; }
            vertices[id] = trimesh::point((float)x, (float)y, (float)z);
            seen[id] = 1;
        }
        for (int i = 0; i < n_nodes; ++i) if (!seen[i]) { std::cerr << "\n**TetMesh Error: Your indices are bad for file " << path << std::endl; return false; }
        return true;
    }
    // TetMesh.cpp:180-228
    bool load_ele(const std::string &filename) {
        const std::string path = filename + ".ele";
        std::ifstream in(path.c_str());
        if (!in) { std::cerr << "\n**TetMesh Error: Could not load " << path << std::endl; return false; }
        std::string line;
        std::getline(in, line);
        int n_tets = 0; { std::stringstream hs(line); hs >> n_tets; }
        tets.resize(n_tets);
        std::vector<int> seen(n_tets, 0);
        bool one_based = false;
        for (int i = 0; i < n_tets; ++i) {
            std::getline(in, line);
            std::stringstream ls(line);
            int id, n[4];
            ls >> id >> n[0] >> n[1] >> n[2] >> n[3];
            if (i == 0 && id == 1) one_based = true;
            if (one_based) { id -= 1; for (int j = 0; j < 4; ++j) n[j] -= 1; }
            if (id < 0 || id >= n_tets) { std::cerr << "\n**TetMesh Error: Your indices are bad for file " << path << std::endl; return false; }
            for (int j = 0; j < 4; ++j) if (n[j] < 0 || n[j] >= (int)vertices.size()) { std::cerr << "\n**TetMesh Error: element " << id << " of " << path << " names a node that does not exist" << std::endl; return false; }
            tets[id] = tet(n[0], n[1], n[2], n[3]);
            seen[id] = 1;
        }
        for (int i = 0; i < n_tets; ++i) if (!seen[i]) { std::cerr << "\n**TetMesh Error: Your indices are bad for file " << path << std::endl; return false; }
        return true;
    }
    // TetMesh.cpp:231-271: faces used by exactly one tet, in the count table's iteration order
    bool need_surface() {
        std::unordered_map<int3, int, int3_hash> count;
        for (size_t t = 0; t < tets.size(); ++t) {
            const int p0 = tets[t].v[0], p1 = tets[t].v[1], p2 = tets[t].v[2], p3 = tets[t].v[3];
            const int3 f[4] = {int3(p0, p1, p3), int3(p0, p2, p1), int3(p0, p3, p2), int3(p1, p2, p3)};
            for (int k = 0; k < 4; ++k) {
                if (count.count(f[k]) == 0) count[f[k]] = 1;
                else count[f[k]] += 1;
            }
        }
        for (std::unordered_map<int3, int, int3_hash>::iterator it = count.begin(); it != count.end(); ++it)
            if (it->second == 1) faces.push_back(trimesh::TriMesh::Face(it->first.orig_v[0], it->first.orig_v[1], it->first.orig_v[2]));
        return true;
    }
};

// DefaultBuilders.hpp:50-304 for the object types a force can be attached to (plane, the five primitives, tetmesh); everything else -> StaticShape
static inline std::shared_ptr<BaseObject> default_build_object(Component &obj) {
    const std::string type = parse::to_lower(obj.type);
    trimesh::xform x_form;
    std::string material = "";
    for (size_t i = 0; i < obj.params.size(); ++i) {
        const std::string tag = parse::to_lower(obj.params[i].tag);
        if (tag == "translate" || tag == "scale" || tag == "rotate") x_form = x_form * obj.params[i].as_xform();
        else if (tag == "material") material = obj.params[i].as_string();
    }
    if (type == "plane") {
        std::shared_ptr<trimesh::TriMesh> tris(new trimesh::TriMesh());
        int width = 10, length = 10;
        double noise = 0.0;
        for (size_t i = 0; i < obj.params.size(); ++i) {
            const std::string tag = parse::to_lower(obj.params[i].tag);
            if (tag == "width") width = obj.params[i].as_int();
            else if (tag == "length") length = obj.params[i].as_int();
            else if (tag == "noise") noise = obj.params[i].as_double();
        }
        if (noise > 0.0) throw std::runtime_error("\n**Scene Error: object \"" + obj.name + "\": <noise> uses trimesh2's random noisify, which this loader does not carry");
        trimesh::make_sym_plane(tris.get(), width, length);
        tris->need_tstrips();
        std::shared_ptr<BaseObject> o(new TriangleMesh(tris, material));
        o->apply_xform(x_form);
        return o;
    }
    // mclscene's primitives (DefaultBuilders.hpp:83-256): triangle meshes like any other -- ForceBuilder attaches forces to them the same way
    if (type == "sphere" || type == "box" || type == "cube" || type == "beam" || type == "cylinder" || type == "torus") {
        std::shared_ptr<trimesh::TriMesh> tris(new trimesh::TriMesh());
        auto par = [&](const char *key) -> const Param * { const Param *hit = 0; for (size_t i = 0; i < obj.params.size(); ++i) if (parse::to_lower(obj.params[i].tag) == key) hit = &obj.params[i]; return hit; };
        if (type == "sphere") {
            double radius = 1.0; trimesh::vec center(0, 0, 0); int tessellation = 1;
            if (const Param *q = par("radius")) radius = q->as_double();
            if (const Param *q = par("center")) center = q->as_vec3();
            if (const Param *q = par("tess")) tessellation = q->as_int();
            trimesh::make_sphere_polar(tris.get(), tessellation, tessellation);
            trimesh::apply_xform(tris.get(), trimesh::xform::scale(radius, radius, radius));
            trimesh::apply_xform(tris.get(), trimesh::xform::trans(center[0], center[1], center[2]));
        } else if (type == "box" || type == "cube") {      // (trimesh2's make_cube alone "is broken": a beam of one chunk, DefaultBuilders.hpp:122-134)
            int tess = 3;
            if (const Param *q = par("tess")) tess = q->as_int();
            trimesh::make_beam(tris.get(), tess, 1);
        } else if (type == "beam") {
            int tess = 3, chunks = 5;
            if (const Param *q = par("tess")) tess = q->as_int();
            if (const Param *q = par("chunks")) chunks = q->as_int();
            trimesh::make_beam(tris.get(), tess, chunks);
        } else if (type == "cylinder") {
            float radius = 1.f; int tess_l = 10, tess_c = 10;
            if (const Param *q = par("tess_l")) tess_l = q->as_int();
            if (const Param *q = par("tess_c")) tess_c = q->as_int();
            if (const Param *q = par("radius")) radius = q->as_float();
            trimesh::make_ccyl(tris.get(), tess_l, tess_c, radius);
        } else {
            int tess_th = 50, tess_ph = 20; float inner_rad = 0.25f;
            if (const Param *q = par("tess_th")) tess_th = q->as_int();
            if (const Param *q = par("tess_ph")) tess_ph = q->as_int();
            if (const Param *q = par("inner_radius")) inner_rad = q->as_float();
            trimesh::make_torus(tris.get(), tess_th, tess_ph, inner_rad, 1.f);
        }
        tris->need_tstrips();
        std::shared_ptr<BaseObject> o(new TriangleMesh(tris, material));
        o->apply_xform(x_form);
        return o;
    }
    if (type == "trimesh") {      // DefaultBuilders.hpp:258-285: read the file, drop unreferenced vertices
        std::string filename = "";
        for (size_t i = 0; i < obj.params.size(); ++i) if (parse::to_lower(obj.params[i].tag) == "file") filename = obj.params[i].as_string();
        if (!filename.size()) throw std::runtime_error("\n**TriangleMesh Error for obj " + obj.name + ": No file specified");
        std::shared_ptr<trimesh::TriMesh> tris(new trimesh::TriMesh());
        std::string why;
        if (!trimesh::read_mesh_file(filename.c_str(), tris.get(), &why)) {
            if (!obj.exists("force")) return std::shared_ptr<BaseObject>(new StaticShape(type, material));      // scenery in a format this loader does not read: parameters only, as before
            throw std::runtime_error("\n**TriangleMesh Error for obj " + obj.name + ": failed to load file " + filename + " (" + why + ")");
        }
        trimesh::remove_unused_vertices(tris.get());
        tris->need_tstrips();
        std::shared_ptr<BaseObject> o(new TriangleMesh(tris, material));
        o->apply_xform(x_form);
        return o;
    }
    if (type == "tetmesh") {
        std::shared_ptr<TetMesh> mesh(new TetMesh(material));
        std::string filename = "";
        for (size_t i = 0; i < obj.params.size(); ++i) if (parse::to_lower(obj.params[i].tag) == "file") filename = obj.params[i].as_string();
        if (!filename.size()) throw std::runtime_error("\n**TetMesh Error for obj " + obj.name + ": No file specified");
        if (!mesh->load(filename)) throw std::runtime_error("\n**TetMesh Error for obj " + obj.name + ": failed to load file " + filename);
        std::shared_ptr<BaseObject> o(mesh);
        o->apply_xform(x_form);
        return o;
    }
    return std::shared_ptr<BaseObject>(new StaticShape(type, material));
}

typedef std::function<std::shared_ptr<BaseObject>(Component &)> BuildObjCallback;

// SceneManager.cpp:37-147 without cameras, lights, materials and the BVH
class SceneManager {
public:
    SceneManager() { createObject = default_build_object; }
    bool load(std::string filename) {
        xml::Node doc;
        if (!xml::load_file(filename, doc)) { std::cerr << "\n**SceneManager::load_xml Error: Unable to load " << filename << std::endl; return false; }
        const std::string xmldir = parse::fileDir(filename);
        const xml::Node *head = xml::find_head(doc, "mclscene");
        std::vector<Component> components;
        for (size_t c = 0; head && c < head->children.size(); ++c) {
            const xml::Node &n = head->children[c];
            const std::string name = n.attribute("name"), type = n.attribute("type");
            if (name.size() == 0 || type.size() == 0) { std::cerr << "\n**SceneManager::load_xml Error: Component \"" << n.name << "\" need a name and type." << std::endl; return false; }
            std::vector<Param> params;
            load_params(params, n);
            for (size_t i = 0; i < params.size(); ++i) {
                const std::string t = parse::to_lower(params[i].tag);
                if (t == "file" || t == "texture") params[i].value = xmldir + params[i].as_string();
            }
            components.push_back(Component(n.name, name, type));
            components.back().params = params;
        }
        for (size_t j = 0; j < components.size(); ++j) {
            if (parse::to_lower(components[j].tag) != "object") continue; // cameras, lights, materials: rendering only
            const std::string name = parse::to_lower(components[j].name);
            std::shared_ptr<BaseObject> obj = createObject(components[j]);
            if (obj) { objects.push_back(obj); objects_map[name] = obj; object_params[name] = components[j].params; }
        }
        return true;
    }
    std::vector<std::shared_ptr<BaseObject> > objects;
    std::unordered_map<std::string, std::shared_ptr<BaseObject> > objects_map;   // lower-case name -> object
    std::unordered_map<std::string, std::vector<Param> > object_params;          // parameters in file order
    BuildObjCallback createObject;
};

} // namespace mcl
