// k2_walk4s.hip -- the 4-family walk with the lane-per-family posterior epilogue (k2_mfma.hpp: k2_prune_mfma4<G, NRT_W, true>,
// k2_epilogue_small_r) for tables of at most 64 root sizes: its own translation unit and its own kernels, so that the other
// walk kernels carry none of its code.  Wave tiles of up to three row tiles (a matrix that goes with so few root sizes is
// at most a few row tiles wide; the launcher falls back to the wave-per-family kernels for anything else).
#include "k2_mfma.hpp"

namespace cafehip {

template <int G, int NRT_W, int PR>
static const void* pick4s()
{
    if constexpr (k2_fits4(G, NRT_W) && NRT_W <= 3) return reinterpret_cast<const void*>(&k2_prune_mfma4<G, NRT_W, PR>);
    else return nullptr;
}

template <int G, int PR>
static const void* pick4s_nrt(int nrt_w)
{
    switch (nrt_w) {
        case 1: return pick4s<G, 1, PR>();
        case 2: return pick4s<G, 2, PR>();
        case 3: return pick4s<G, 3, PR>();
    }
    return nullptr;
}

template <int PR>
static const void* pick4s_g(int G, int nrt_w)
{
    switch (G) {
        case 1: return pick4s_nrt<1, PR>(nrt_w);
        case 2: return pick4s_nrt<2, PR>(nrt_w);
        case 3: return pick4s_nrt<3, PR>(nrt_w);
        case 4: return pick4s_nrt<4, PR>(nrt_w);
        case 5: return pick4s_nrt<5, PR>(nrt_w);
        case 6: return pick4s_nrt<6, PR>(nrt_w);
        case 7: return pick4s_nrt<7, PR>(nrt_w);
    }
    return nullptr;
}

// (PR = 2, up to 128 root sizes, was measured too: configs[1] walk 56.6 -> 57.1 us -- beyond 64 root sizes the wave form stays;
// profiles/r06/small_r_epilogue_ab.txt)
const void* k2_mfma4_small_r_kernel(int G, int nrt_w, int pr) { return pr == 1 ? pick4s_g<1>(G, nrt_w) : nullptr; }

}  // namespace cafehip
