// CollisionSphere.hpp -- forwarding header: the reference keeps CollisionSphere in
// deps/admm-elastic-sca/src/collision/CollisionSphere.hpp; callers include it by that name
// (src/ForceBuilder.hpp:23-26, samples/*.cpp).  The mirror declares every force class in Force.hpp.
#pragma once
#include "Force.hpp"
