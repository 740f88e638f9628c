#!/usr/bin/env python3
"""Generate the scene-ingest fixtures tests/golden/scene_*.npz.

Runs in the BUILD container only: it drives oracle/_ref/libscene_ref.so, i.e. the
reference's own SimContext + ForceBuilder + mclscene loader compiled from
/root/reference by `make -C oracle scene_ref` (see oracle/ref_scene_shim.cpp).
One child process per scene (the reference's ForceBuilder keeps static state).

Inputs  : tests/golden/scenes/<sample>/*.xml (+ .node/.ele): the reference's shipped sample
          data files, and tests/golden/scenes/custom/*.xml authored for these tests.
Outputs : per scene, what the reference's loader produced --
          x, m (3 per node), dt, iters, per-force kind / node ids / parameters, explicit forces
          (direction, wind face list), the object iteration order, the surface faces of
          every dynamic object -- and a short trajectory after the sample's own setup steps
          (anchors / wind / cylinder collision, restated in the shim from samples/*/*.cpp).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "..", "oracle", "_ref", "libscene_ref.so")

# name -> (xml, setup, frames)
SCENES = {
    "poordillo": ("scenes/poordillo/poordillo.xml", "none", 2),
    "bunnyexpand": ("scenes/bunnyexpand/bunnyexpand.xml", "scale1.3", 2),
    "windyflag": ("scenes/windyflag/cloth.xml", "flag", 3),
    "windyflag_nowind": ("scenes/windyflag/cloth.xml", "flag_nowind", 3),
    "plinko": ("scenes/plinkopony/plinko.xml", "plinko", 3),
    "two_bodies": ("scenes/custom/two_bodies.xml", "none", 0),
    "uniform_nh": ("scenes/custom/uniform_nh.xml", "none", 0),
    "shapes": ("scenes/custom/shapes.xml", "none", 2),      # round 5: sphere / box / beam / cylinder / torus as dynamic objects
    "objmesh": ("scenes/custom/objmesh.xml", "none", 2),    # round 5: a "trimesh" object from a Wavefront OBJ file
    "plymesh": ("scenes/custom/plymesh.xml", "none", 2),    # round 5: "trimesh" objects from PLY files (ascii, binary little / big endian) and an OFF file
}
DYNAMIC = ("bunny", "dillo", "horse", "cloth1", "sheet", "a", "b", "ball", "crate", "girder", "can", "ring", "patch", "pa", "ple", "pbe", "poff")


def child(name):
    xml, setup, frames = SCENES[name]
    L = C.CDLL(LIB)
    L.refscene_load.restype = C.c_void_p
    L.refscene_load.argtypes = [C.c_char_p]
    for f in ("refscene_initialize", "refscene_step", "refscene_dof", "refscene_n_forces", "refscene_n_explicit", "refscene_add_cylinder_collision"):
        getattr(L, f).argtypes = [C.c_void_p]
    dp = np.ctypeslib.ndpointer(np.float64, flags="C")
    ip = np.ctypeslib.ndpointer(np.int32, flags="C")
    L.refscene_get_x.argtypes = [C.c_void_p, dp]
    L.refscene_get_m.argtypes = [C.c_void_p, dp]
    L.refscene_set_x.argtypes = [C.c_void_p, dp]
    L.refscene_settings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.refscene_get_explicit.argtypes = [C.c_void_p, C.c_int, dp]
    L.refscene_get_force.argtypes = [C.c_void_p, C.c_int, ip, dp]
    L.refscene_object_order.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.refscene_n_faces.argtypes = [C.c_void_p, C.c_char_p]
    L.refscene_n_vertices.argtypes = [C.c_void_p, C.c_char_p]
    L.refscene_get_faces.argtypes = [C.c_void_p, C.c_char_p, ip]
    L.refscene_wind_size.argtypes = [C.c_void_p, C.c_int]
    L.refscene_get_wind.argtypes = [C.c_void_p, C.c_int, ip]
    L.refscene_add_static_anchor.argtypes = [C.c_void_p, C.c_int]
    L.refscene_add_wind.argtypes = [C.c_void_p, dp]
    L.refscene_get_cylinders.argtypes = [C.c_void_p, C.c_int, dp]

    h = L.refscene_load(os.path.join(HERE, xml).encode())
    assert h, "load failed"
    out = {}
    dof = L.refscene_dof(h)
    x = np.zeros(dof); m = np.zeros(dof)
    L.refscene_get_x(h, x); L.refscene_get_m(h, m)
    dt = C.c_double(); it = C.c_int()
    L.refscene_settings(h, C.byref(dt), C.byref(it))
    out.update(x=x.copy(), m=m, dt=dt.value, iters=it.value)
    buf = C.create_string_buffer(1 << 16)
    assert L.refscene_object_order(h, buf, len(buf)) >= 0
    names = [s for s in buf.value.decode().split("\n") if s]
    out["object_order"] = np.array(names)
    for nm in names:
        # static scenery is tessellated by mclscene; only dynamic meshes are part of the contract
        if nm in DYNAMIC:
            nf = L.refscene_n_faces(h, nm.encode())
            out["nverts_" + nm] = L.refscene_n_vertices(h, nm.encode())
            faces = np.zeros(3 * nf, np.int32)
            L.refscene_get_faces(h, nm.encode(), faces)
            out["faces_" + nm] = faces.reshape(-1, 3)

    def dump_forces(tag):
        n = L.refscene_n_forces(h)
        kinds = np.zeros(n, np.int32); idx = np.zeros((n, 4), np.int32); par = np.zeros((n, 4))
        for i in range(n):
            a = np.zeros(4, np.int32); p = np.zeros(4)
            kinds[i] = L.refscene_get_force(h, i, a, p)
            idx[i] = a; par[i] = p
        ne = L.refscene_n_explicit(h)
        dirs = np.zeros((ne, 3)); wsz = np.zeros(ne, np.int32); wind = []
        for i in range(ne):
            d = np.zeros(3); L.refscene_get_explicit(h, i, d); dirs[i] = d
            wsz[i] = L.refscene_wind_size(h, i)
            if wsz[i]:
                w = np.zeros(wsz[i], np.int32); L.refscene_get_wind(h, i, w); wind.append(w)
        out[tag + "kinds"] = kinds; out[tag + "idx"] = idx; out[tag + "par"] = par
        out[tag + "explicit_dir"] = dirs; out[tag + "wind_size"] = wsz
        out[tag + "wind"] = np.concatenate(wind) if wind else np.zeros(0, np.int32)

    dump_forces("load_")          # straight after load(): what ForceBuilder made
    # the sample main's setup() between load() and initialize()
    if setup in ("flag", "flag_nowind"):
        L.refscene_add_static_anchor(h, 0)
        L.refscene_add_static_anchor(h, 20)
        if setup == "flag":
            L.refscene_add_wind(h, np.array([10.0, 0.0, 2.0]))
    elif setup == "plinko":
        nc = L.refscene_add_cylinder_collision(h)
        cyl = np.zeros((nc, 4))
        L.refscene_get_cylinders(h, L.refscene_n_forces(h) - 1, cyl)
        out["cylinders"] = cyl
    assert L.refscene_initialize(h)
    dump_forces("init_")          # after SimContext::initialize(): + gravity / anchors / wind
    if setup == "scale1.3":       # deterministic stand-in for bunnyexpand.cpp's random scramble (after initialize)
        L.refscene_get_x(h, x)
        L.refscene_set_x(h, np.ascontiguousarray(x * 1.3))
    traj = []
    for _ in range(frames):
        L.refscene_step(h)
        L.refscene_get_x(h, x)
        traj.append(x.copy())
    out["traj"] = np.array(traj).reshape(frames, dof)
    out["setup"] = setup
    np.savez_compressed(os.path.join(HERE, "scene_%s.npz" % name), **out)
    print(name, "dof", dof, "forces", L.refscene_n_forces(h), "explicit", L.refscene_n_explicit(h), "objects", names[:4], flush=True)
    os._exit(0)  # skip the reference's static destructors


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        if not os.path.exists(LIB):
            sys.exit("build oracle/_ref/libscene_ref.so first: make -C oracle scene_ref")
        for nm in SCENES:
            env = dict(os.environ, OMP_NUM_THREADS="1")  # WindForce's scatter order is thread dependent
            r = subprocess.run([sys.executable, os.path.abspath(__file__), nm], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            lines = r.stdout.strip().splitlines()
            print(lines[-1] if lines else "(no output)")
            if r.returncode != 0:
                print(r.stdout)
                sys.exit("scene %s failed" % nm)
