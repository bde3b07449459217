// k2_walk4.hip -- the family walk on v_mfma_f64_4x4x4_4b (k2_mfma.hpp: k2_prune_mfma4<G, NRT_W>): family tiles of any
// multiple of 4, so that a small table fills the chip evenly.  Only the (G, NRT_W) wave tiles that compile without
// scratch spills are instantiated (k2_fits4; tools/k2_regs.py).
#include "k2_mfma.hpp"

namespace cafehip {

#ifndef CAFE_K2_OBJ
#define CAFE_K2_OBJ false
#define CAFE_K2_GETTER4 k2_mfma4_kernel
#endif

template <int G, int NRT_W>
static const void* pick4()
{
    if constexpr (k2_fits4(G, NRT_W)) return reinterpret_cast<const void*>(&k2_prune_mfma4<G, NRT_W, 0, CAFE_K2_OBJ>);
    else return nullptr;
}

template <int G>
static const void* pick4_nrt(int nrt_w)
{
    switch (nrt_w) {
        case 1: return pick4<G, 1>();
        case 2: return pick4<G, 2>();
        case 3: return pick4<G, 3>();
        case 4: return pick4<G, 4>();
        case 5: return pick4<G, 5>();
        case 6: return pick4<G, 6>();
        case 7: return pick4<G, 7>();
    }
    return nullptr;
}

const void* CAFE_K2_GETTER4(int G, int nrt_w)
{
    switch (G) {
        case 1: return pick4_nrt<1>(nrt_w);
        case 2: return pick4_nrt<2>(nrt_w);
        case 3: return pick4_nrt<3>(nrt_w);
        case 4: return pick4_nrt<4>(nrt_w);
        case 5: return pick4_nrt<5>(nrt_w);
        case 6: return pick4_nrt<6>(nrt_w);
        case 7: return pick4_nrt<7>(nrt_w);
        case 8: return pick4_nrt<8>(nrt_w);
    }
    return nullptr;
}

}  // namespace cafehip
