/*
 * admm_kinds.h -- force-kind enumeration and per-kind array shapes shared by
 * the HIP library (include/admm_hip.h), the CPU oracle (oracle/admm_oracle.h)
 * and the reference shim (oracle/ref_shim.cpp).
 *
 * Each kind mirrors one admm::Force subclass of the reference; the number of
 * nodes, D rows and parameters per element follow that class's
 * get_selector()/project():
 *
 *   kind          reference class (file:line)                         nodes rows
 *   ANCHOR        StaticAnchor / MovingAnchor  AnchorForce.cpp:31-89     1    3
 *   SPRING        Spring                       Force.cpp:29-71           2    3
 *   TET_LINEAR    LinearTetStrain              TetForce.cpp:112-153      4    9
 *   TET_VOLUME    TetVolume                    TetForce.cpp:160-210      4    9
 *   TET_NH        HyperElasticTet type 0       TetForce.cpp:303-364      4    9
 *   TET_STVK      HyperElasticTet type 1       TetForce.cpp:303-364      4    9
 *   TRI_STRAIN    LimitedTriangleStrain        TriangleForce.cpp:29-113  3    6
 *   BEND          BendForce                    BendForce.cpp:26-161      4    9
 *   COLLISION     CollisionForce (one element per node)  CollisionForce.cpp:27-70  1    3
 *   TRI_AREA      TriArea                      TriangleForce.cpp:257-295 3    6
 *   TRI_FUNG      FungTriangle                 TriangleForce.cpp:120-249 3    6
 *   GENERIC       any user-written admm::Force subclass  Force.hpp:37-57   (rows and nodes per element as its
 *                 get_selector says; project() runs in the caller's hook: admm_hip_add_generic_batch)
 *
 * "rows" are the compact rows of D/u/z per element.  (The reference gives
 * every tet 36 rows of which 27 are structurally zero -- TetForce.cpp:61 vs
 * :313-317 -- see DESIGN.md "row layout".)
 */
#ifndef ADMM_KINDS_H
#define ADMM_KINDS_H

#ifdef __cplusplus
extern "C" {
#endif

enum admm_kind {
    ADMM_KIND_ANCHOR     = 0,
    ADMM_KIND_SPRING     = 1,
    ADMM_KIND_TET_LINEAR = 2,
    ADMM_KIND_TET_VOLUME = 3,
    ADMM_KIND_TET_NH     = 4,
    ADMM_KIND_TET_STVK   = 5,
    ADMM_KIND_TRI_STRAIN = 6,
    ADMM_KIND_BEND       = 7,
    ADMM_KIND_COLLISION  = 8,
    ADMM_KIND_TRI_AREA   = 9,
    ADMM_KIND_TRI_FUNG   = 10,
    ADMM_KIND_COUNT      = 11,  /* built-in kinds (size of the tables below) */
    ADMM_KIND_GENERIC    = 11   /* a run of user-defined admm::Force subclasses: selector rows given as triplets, project() runs in
                                   the caller's hook on the host (admm_hip_add_generic_batch); no table entries */
};

/* nodes per element */
static const int ADMM_KIND_NODES[ADMM_KIND_COUNT]  = { 1, 2, 4, 4, 4, 4, 3, 4, 1, 3, 3 };
/* compact D rows per element */
static const int ADMM_KIND_ROWS[ADMM_KIND_COUNT]   = { 3, 3, 9, 9, 9, 9, 6, 9, 3, 6, 6 };
/* doubles of constructor parameters per element (see admm_hip_add_batch) */
static const int ADMM_KIND_PARAMS[ADMM_KIND_COUNT] = { 2, 1, 1, 3, 3, 3, 4, 1, 1, 4, 3 };
/* doubles of persistent warm-start state per element */
static const int ADMM_KIND_STATE[ADMM_KIND_COUNT]  = { 0, 0, 0, 0, 4, 4, 0, 0, 0, 0, 4 };

/*
 * params layout per kind (doubles, element-major [n_elems][ADMM_KIND_PARAMS]):
 *   ANCHOR      { use_weight (<=0 -> 1000.f, AnchorForce.hpp:57-60), active (1/0; MovingAnchor only) }
 *               anchor target positions are passed separately (rest = NULL -> x at initialize)
 *   SPRING      { stiffness }
 *   TET_LINEAR  { stiffness }
 *   TET_VOLUME  { stiffness, limit_min, limit_max }
 *   TET_NH/STVK { mu, lambda, max_iterations }
 *   TRI_STRAIN  { stiffness, limit_min, limit_max, strain_limiting (1/0) }
 *   BEND        { stiffness }
 *   COLLISION   { use_weight }   (CollisionForce.hpp:33: default 32); the reference's single
 *               CollisionForce over all nodes is one batch with one element per node, in node
 *               order; the shapes are set with admm_hip_set_collision_shapes
 *   TRI_AREA    { stiffness, iters, limit_min, limit_max }   (TriangleForce.hpp:126-133)
 *   TRI_FUNG    { mu, limit_min, limit_max }                  (TriangleForce.hpp:106-124; L-BFGS maxIter 10, gradTol 1e-6)
 * state layout (TET_NH/STVK): { last_prox_result[3], init_hess }  (TetForce.hpp:146, meta.h:33)
 *              (TRI_FUNG):    { -, -, -, init_hess }  the solver's persisted Hessian guess; no warm start
 */

/* analytic collision shapes (deps/admm-elastic-sca/src/collision/), tested in list order */
enum admm_shape {
    ADMM_SHAPE_FLOOR    = 0,   /* params { -, cy, -, - }        CollisionFloor.hpp:51-58    */
    ADMM_SHAPE_SPHERE   = 1,   /* params { cx, cy, cz, radius } CollisionSphere.hpp:50-66   */
    ADMM_SHAPE_CYLINDER = 2    /* params { cx, cy, -, radius }  z-axis, CollisionCylinder.hpp:48-66 */
};
#define ADMM_MAX_SHAPES 64

/* explicit (pre-step) forces, applied to v in list order once per frame (ExplicitForce.cpp) */
enum admm_explicit {
    ADMM_EXPLICIT_CONST = 0,   /* v += dt*dir on all nodes, or on an index subset  ExplicitForce.cpp:29-39 */
    ADMM_EXPLICIT_WIND  = 1    /* per-triangle aerodynamic drag                    ExplicitForce.cpp:42-98 */
};

#ifdef __cplusplus
}
#endif
#endif
