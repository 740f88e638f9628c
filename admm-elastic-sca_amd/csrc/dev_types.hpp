// dev_types.hpp -- plain records shared by the kernels (kernels_global.hpp, kernels_local.hpp) and the host-only translation units
// (ctx.hpp): no device code in here, so that a host compile can include it.
#pragma once
#include <stdint.h>
#include "../../include/admm_kinds.h"

namespace admm_dev {

// elements per workgroup of the local-step kernels = one 64-lane wave (kernels_local.hpp; the tet kernels may use fewer lanes: Batch::tpb)
#ifndef ADMM_LOCAL_BLOCK
#define ADMM_LOCAL_BLOCK 64
#endif
constexpr int LOCAL_BLOCK = ADMM_LOCAL_BLOCK;

// explicit all-node forces of a frame (ExplicitForce.cpp:29-39), passed to prologue_kernel by value
constexpr int MAX_GRAV = 4;
struct Gravity { int n; double g[MAX_GRAV][3]; };

// analytic collision shapes of a CollisionForce (collision/*.hpp), tested in list order by project_collision_block
struct ShapeTable { int n; int type[ADMM_MAX_SHAPES]; double par[ADMM_MAX_SHAPES][4]; };

// One work item of a sweep launch with everything the block needs to start, in one 64-byte record
// (one scalar load instead of an index load followed by six dependent per-supernode loads).
struct __attribute__((aligned(16))) SweepItem {
    int s, part;                 // supernode; 64-row tile (forward) or column chunk (backward)
    int k, r;                    // columns, below-diagonal rows
    int first, pad;              // first column in factor order
    long long panel_off, front_off, slot_off, rows_off;
    long long pad2;              // (profile build -DADMM_SWEEP_PROFILE: the workgroup's slot in the stamp buffer, -1 = none)
};
static_assert(sizeof(SweepItem) == 64, "SweepItem is one 64-byte record");

} // namespace admm_dev
