// k2_walk16.hip -- the family walk on v_mfma_f64_16x16x4 (k2_mfma.hpp: k2_prune_mfma<NFT_W, NRT_W>): pruning of all
// families + posterior == compute_tree_likelihoods + compute_posterior, cafe/cafe_tree.c:191-323, cafe/lambda.cpp:657-689.
// Only the (NFT_W, NRT_W) wave tiles within the register budget are instantiated (k2_fits16).
#include "k2_mfma.hpp"

namespace cafehip {

#ifndef CAFE_K2_OBJ
#define CAFE_K2_OBJ false
#define CAFE_K2_GETTER16 k2_mfma16_kernel
#endif

template <int NFT_W, int NRT_W>
static const void* pick16()
{
    if constexpr (k2_fits16(NFT_W, NRT_W)) return reinterpret_cast<const void*>(&k2_prune_mfma<NFT_W, NRT_W, CAFE_K2_OBJ>);
    else return nullptr;
}

const void* CAFE_K2_GETTER16(int nft_w, int nrt_w)
{
#define CAFE_M16(F, N) if (nft_w == F && nrt_w == N) return pick16<F, N>();
    CAFE_M16(1, 1) CAFE_M16(1, 2) CAFE_M16(1, 3) CAFE_M16(1, 4) CAFE_M16(1, 5) CAFE_M16(1, 6) CAFE_M16(1, 7)
    CAFE_M16(2, 1) CAFE_M16(2, 2) CAFE_M16(2, 3) CAFE_M16(2, 4)
#undef CAFE_M16
    return nullptr;
}

}  // namespace cafehip
