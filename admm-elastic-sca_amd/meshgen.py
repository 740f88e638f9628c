"""Synthetic scene generators for the configurations of BASELINE.md section 4 /
SURVEY.md section 8(d).  Pure numpy, deterministic, seed-free.

`bar()` is the "synthetic NH bar" of config 4: an nx*ny*nz grid of cubes with
spacing h, node id = i + (nx+1)*(j + (ny+1)*k), each cube split into 6 Kuhn
tets (one per permutation of the axes, order 012,021,102,120,201,210; vertices
= the cube corner (i,j,k) followed by successive +1 steps along the permuted
axes; last two vertices swapped when the signed volume is negative), tets
emitted cube-major (i fastest).  32x32x163 gives 1 001 472 tets / 178 596 nodes.
"""
import itertools

import numpy as np

KUHN_PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def bar(nx, ny, nz, h=0.05):
    """Returns (x [n,3] float64, tets [m,4] int32)."""
    gx, gy, gz = nx + 1, ny + 1, nz + 1
    k, j, i = np.meshgrid(np.arange(gz), np.arange(gy), np.arange(gx), indexing="ij")
    x = np.stack([i.ravel() * h, j.ravel() * h, k.ravel() * h], axis=1).astype(np.float64)

    ck, cj, ci = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    corner = np.stack([ci.ravel(), cj.ravel(), ck.ravel()], axis=1)  # cube-major, i fastest
    ncube = corner.shape[0]
    tets = np.empty((ncube, 6, 4), dtype=np.int64)
    strides = np.array([1, gx, gx * gy], dtype=np.int64)
    for p, perm in enumerate(KUHN_PERMS):
        c = corner.copy()
        tets[:, p, 0] = c @ strides
        for s, ax in enumerate(perm):
            c[:, ax] += 1
            tets[:, p, s + 1] = c @ strides
    tets = tets.reshape(-1, 4)
    # orientation: swap the last two vertices where the signed volume is negative
    v0, v1, v2, v3 = (x[tets[:, a]] for a in range(4))
    vol = np.einsum("ij,ij->i", np.cross(v1 - v0, v2 - v0), v3 - v0)
    neg = vol < 0
    tets[neg, 2], tets[neg, 3] = tets[neg, 3].copy(), tets[neg, 2].copy()
    return x, tets.astype(np.int32)


def lumped_tet_mass(x, tets, density):
    """rho * vol / 4 to each corner (ForceBuilder density-weighted mass,
    reference src/ForceBuilder.hpp:191-303); returns [n] per-node mass."""
    v0, v1, v2, v3 = (x[tets[:, a]] for a in range(4))
    vol = np.abs(np.einsum("ij,ij->i", np.cross(v1 - v0, v2 - v0), v3 - v0)) / 6.0
    m = np.zeros(x.shape[0])
    np.add.at(m, tets.ravel(), np.repeat(density * vol / 4.0, 4))
    return m


def bar_anchor_nodes(nx, ny):
    """node ids of the k = 0 face."""
    return np.arange((nx + 1) * (ny + 1), dtype=np.int32)


def sym_plane(w, l, size=1.0):
    """A (w x l)-cell 'symmetric plane' cloth: every cell has a centre vertex
    and 4 triangles (4*w*l tris, (w+1)(l+1)+w*l nodes) in the xz-plane, as
    trimesh2's make_sym_plane lays it out (reference
    deps/mclscene/deps/trimesh2/include/TriMeshBuilder.h:116-171).  Geometry is
    generated in float64 here (the reference rounds vertices to float)."""
    nv = (w + 1) * (l + 1)
    ii, jj = np.meshgrid(np.arange(w + 1), np.arange(l + 1), indexing="xy")
    gx = (ii.ravel() / w - 0.5) * size
    gz = (jj.ravel() / l - 0.5) * size * l / w
    ci, cj = np.meshgrid(np.arange(w), np.arange(l), indexing="xy")
    cx = ((ci.ravel() + 0.5) / w - 0.5) * size
    cz = ((cj.ravel() + 0.5) / l - 0.5) * size * l / w
    x = np.zeros((nv + w * l, 3))
    x[:nv, 0] = gx; x[:nv, 2] = gz
    x[nv:, 0] = cx; x[nv:, 2] = cz
    tris = []
    for j, i in itertools.product(range(l), range(w)):
        a = i + (w + 1) * j; b = a + 1; c = a + (w + 1); d = c + 1; m = nv + i + w * j
        tris += [(a, b, m), (b, d, m), (d, c, m), (c, a, m)]
    return x, np.array(tris, dtype=np.int32)


def bend_hinges(tris):
    """One hinge (i0, i1, i2, i3) per interior edge: i2,i3 the shared edge,
    i0/i1 the opposite vertices (BendForce rows are x0-x2, x3-x2, x1-x2;
    reference deps/admm-elastic-sca/src/system/BendForce.cpp:58-118)."""
    edge = {}
    hinges = []
    for t, (a, b, c) in enumerate(tris):
        for (p, q, o) in ((a, b, c), (b, c, a), (c, a, b)):
            key = (min(p, q), max(p, q))
            if key in edge:
                o2 = edge[key]
                hinges.append((o2, o, key[0], key[1]))
            else:
                edge[key] = o
    return np.array(hinges, dtype=np.int32).reshape(-1, 4)
