// k2c_tables.hip -- k2c_nodes (k2_mfma.hpp): factor tables of compressed subtrees, one launch per level.  No reference
// counterpart; values of cafe/cafe_tree.c:191-323.
#include "k2_mfma.hpp"

namespace cafehip {

const void* k2c_kernel(int nft_w, int nrt_w, bool batch_gathers, bool pair)
{
    if (pair) {
        if (nft_w == 1 && nrt_w == 2) return batch_gathers ? reinterpret_cast<const void*>(&k2c_nodes<1, 2, true, true>) : reinterpret_cast<const void*>(&k2c_nodes<1, 2, false, true>);
        return nullptr;
    }
    if (nft_w == 1 && nrt_w == 1) return batch_gathers ? reinterpret_cast<const void*>(&k2c_nodes<1, 1, true>) : reinterpret_cast<const void*>(&k2c_nodes<1, 1, false>);
    if (nft_w == 1 && nrt_w == 2) return batch_gathers ? reinterpret_cast<const void*>(&k2c_nodes<1, 2, true>) : reinterpret_cast<const void*>(&k2c_nodes<1, 2, false>);
    return nullptr;
}

}  // namespace cafehip
