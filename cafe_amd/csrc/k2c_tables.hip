// k2c_tables.hip -- k2c_nodes (k2_mfma.hpp): factor tables of compressed subtrees, one launch per level.  No reference
// counterpart; values of cafe/cafe_tree.c:191-323.
#include "k2_mfma.hpp"

namespace cafehip {

const void* k2c_kernel(int nft_w, int nrt_w, int kpf, int kmax)
{
    // kpf: k-steps of the matrix operand requested up front (ring of that depth, straight-line product over <= kmax
    // k-steps); kpf 0 = the walk's short ring; kpf -1 = the short ring with the child columns gathered in one batch
#define CAFE_K2C_PF(P, M) if (nft_w == 1 && nrt_w == 1 && kpf == P && kmax == M) return reinterpret_cast<const void*>(&k2c_nodes<1, 1, P, M, true>);
    CAFE_K2C_PF(40, 40) CAFE_K2C_PF(8, 64) CAFE_K2C_PF(12, 64) CAFE_K2C_PF(16, 64) CAFE_K2C_PF(24, 64) CAFE_K2C_PF(32, 64)
#undef CAFE_K2C_PF
    if (nft_w == 1 && nrt_w == 1 && kpf == -1) return reinterpret_cast<const void*>(&k2c_nodes<1, 1, 0, 0, true>);
    if (nft_w == 1 && nrt_w == 1 && kpf == 0) return reinterpret_cast<const void*>(&k2c_nodes<1, 1, 0, 0, false>);
    if (nft_w == 1 && nrt_w == 2 && kpf == 0) return reinterpret_cast<const void*>(&k2c_nodes<1, 2, 0, 0, false>);
    return nullptr;
}

}  // namespace cafehip
