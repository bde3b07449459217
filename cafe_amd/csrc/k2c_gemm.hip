// k2c_gemm.hip -- k2c_gemm (k2c_gemm.hpp): factor tables of compressed subtrees as a tiled GEMM, one launch per level.  No
// reference counterpart; values of cafe/cafe_tree.c:191-323.
// Built with -mllvm -amdgpu-mfma-vgpr-form=1 (cafe_amd/build.py): with a 256-register budget the compiler otherwise keeps
// the accumulators in AGPRs inside the chunk loop and in VGPRs across its back edge -- 128 v_accvgpr moves per chunk.
#include "k2c_gemm.hpp"

namespace cafehip {

#define K2G_INST(NST, NRT, GS, MAXT) \
    if (nst == NST && nrt_w == NRT && gs == GS && max_threads == MAXT) return reinterpret_cast<const void*>(&k2c_gemm<NST, NRT, GS, MAXT>);

const void* k2c_gemm_kernel(int nst, int nrt_w, int gs, int max_threads)
{
    K2G_INST(1, 1, 1, 512) K2G_INST(1, 1, 2, 512) K2G_INST(1, 1, 4, 512)
    K2G_INST(1, 2, 1, 512) K2G_INST(1, 2, 2, 512) K2G_INST(1, 2, 4, 512)
    K2G_INST(2, 1, 1, 512) K2G_INST(2, 1, 2, 512) K2G_INST(2, 1, 4, 512)
    K2G_INST(2, 2, 1, 512) K2G_INST(2, 2, 2, 512) K2G_INST(2, 2, 4, 512)
    K2G_INST(4, 1, 1, 512) K2G_INST(4, 1, 2, 512) K2G_INST(4, 1, 4, 512)
    K2G_INST(4, 2, 1, 512) K2G_INST(4, 2, 2, 512) K2G_INST(4, 2, 4, 512)
    // more than 8 waves (one row tile per wave above 8 row tiles, pairs above 16): 128 registers per lane
    K2G_INST(1, 1, 1, 1024) K2G_INST(1, 2, 1, 1024) K2G_INST(2, 1, 1, 1024) K2G_INST(2, 2, 1, 1024)
    K2G_INST(4, 1, 1, 1024) K2G_INST(4, 1, 2, 1024) K2G_INST(4, 2, 1, 1024) K2G_INST(4, 2, 2, 1024)
    return nullptr;
}
#undef K2G_INST

}  // namespace cafehip
