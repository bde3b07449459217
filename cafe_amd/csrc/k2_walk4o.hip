// k2_walk4o.hip -- the objective-only instantiations of the 4-family walk (k2_mfma.hpp: k2_prune_mfma4<G, NRT_W, 0, true>): the
// same table of wave tiles as k2_walk4.hip, compiled as its own unit.
#define CAFE_K2_OBJ true
#define CAFE_K2_GETTER4 k2_mfma4_objective_kernel
#include "k2_walk4.hip"
