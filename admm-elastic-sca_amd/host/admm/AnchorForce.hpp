// AnchorForce.hpp -- forwarding header: the reference keeps StaticAnchor, MovingAnchor, ControlPoint, helper::smooth_move / linear_move in
// deps/admm-elastic-sca/src/system/AnchorForce.hpp; callers include it by that name
// (src/ForceBuilder.hpp:23-26, samples/*.cpp).  The mirror declares every force class in Force.hpp.
#pragma once
#include "Force.hpp"
