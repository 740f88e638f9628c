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
    ADMM_KIND_COUNT      = 8
};

/* nodes per element */
static const int ADMM_KIND_NODES[ADMM_KIND_COUNT]  = { 1, 2, 4, 4, 4, 4, 3, 4 };
/* compact D rows per element */
static const int ADMM_KIND_ROWS[ADMM_KIND_COUNT]   = { 3, 3, 9, 9, 9, 9, 6, 9 };
/* doubles of constructor parameters per element (see admm_hip_add_batch) */
static const int ADMM_KIND_PARAMS[ADMM_KIND_COUNT] = { 2, 1, 1, 3, 3, 3, 4, 1 };
/* doubles of persistent warm-start state per element */
static const int ADMM_KIND_STATE[ADMM_KIND_COUNT]  = { 0, 0, 0, 0, 4, 4, 0, 0 };

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
 * state layout (TET_NH/STVK): { last_prox_result[3], init_hess }  (TetForce.hpp:146, meta.h:33)
 */

#ifdef __cplusplus
}
#endif
#endif
