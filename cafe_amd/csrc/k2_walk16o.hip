// k2_walk16o.hip -- the objective-only instantiations of the 16-family walk (k2_mfma.hpp: k2_prune_mfma<NFT_W, NRT_W, true>): the
// same table of wave tiles as k2_walk16.hip, compiled as its own unit.
#define CAFE_K2_OBJ true
#define CAFE_K2_GETTER16 k2_mfma16_objective_kernel
#include "k2_walk16.hip"
