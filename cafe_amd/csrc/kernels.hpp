// kernels.hpp -- what the kernel translation units hand to the context (cafehip.hip): the ADDRESS of a kernel
// instantiation (or nullptr when that shape is not built), launched with hipLaunchKernel and one argument struct
// (device_types.hpp).  Splitting the instantiations over several units keeps a rebuild after a kernel edit to the
// unit that holds the kernel, and the units compile in parallel (cafe_amd/build.py).
#pragma once
#include "device_types.hpp"

namespace cafehip {

// k1_matrices.hip
const void* k1_kernel(bool use_lds, bool product_form);   // k1_build_matrices<USE_LDS, PRODUCT_FORM>(K1Args)
const void* k1_rb_kernel();                               // k1_build_matrices_rb(K1Args)
const void* k3_then_k1_rb_kernel();                       // k3_score_then_k1_rb(K3K1Args): score blocks, then matrix-build blocks
int k1_rb_columns();                                      // columns per thread of the register-blocked kernel
int k1_rb_bpad();                                         // zeros staged in front of every B row
const void* k1e_fold_kernel();                            // k1e_fold_error(FoldArgs)

// k2_walk16.hip / k2_walk4.hip: the family walk on the matrix cores (k2_mfma.hpp), K2MfmaArgs
constexpr int kMaxTiles16 = 8;      // NFT_W * NRT_W accumulator tiles per wave (16x16x4 shape)
constexpr bool k2_fits16(int nft_w, int nrt_w) { return nft_w >= 1 && nft_w <= 2 && nrt_w >= 1 && nrt_w <= 7 && nft_w * nrt_w <= kMaxTiles16; }
// (G, NRT_W) wave tiles of the 4-family kernel that compile without scratch spills at 2 waves per SIMD (256 registers
// per lane; checked with tools/k2_regs.py after every kernel change)
constexpr bool k2_fits4(int G, int nrt_w) { return G >= 1 && G <= 7 && nrt_w >= 1 && nrt_w <= 7 && G * nrt_w <= 18; }   // (8, 1): 20 B of scratch
const void* k2_mfma16_kernel(int nft_w, int nrt_w);
const void* k2_mfma4_kernel(int G, int nrt_w);
// k2_walk16o.hip / k2_walk4o.hip: the same kernels without the batch mode's and the unfolded error model's code, for launches
// with col_max == NULL and (err == NULL or PTfold != NULL): every objective evaluation
const void* k2_mfma16_objective_kernel(int nft_w, int nrt_w);
const void* k2_mfma4_objective_kernel(int G, int nrt_w);
// k2_walk4s.hip: the same walk with the lane-per-family posterior epilogue, R <= 64 and NF <= 96 only (NULL: not instantiated)
const void* k2_mfma4_small_r_kernel(int G, int nrt_w, int pr = 1);   // pr: R <= 64 * pr (1, 2)

// k2c_tables.hip: factor tables of compressed subtrees, K2cArgs (batch_gathers: the child columns of a state in one batch)
const void* k2c_kernel(int nft_w, int nrt_w, bool batch_gathers, bool pair);

// k2c_gemm.hip: the same tables as a tiled GEMM (k2c_gemm.hpp): nst state tiles per workgroup, nrt_w row tiles per wave, gs gather
// slots per thread (1, 2, 4)
const void* k2c_gemm_kernel(int nst, int nrt_w, int gs, int max_threads);   // max_threads: 512 or 1024

// k_misc.hip
const void* k2_v1_kernel(int nf, bool reference_arithmetic = false);   // k2_prune_v1<NF, REF>(K2Args), NF in {1, 2, 4, 8, 16}
const void* k3_kernel(bool host_out);      // k3_score<HOST_OUT>(K3Args)
const void* k3x_kernel();                  // k3_score_x(K3xArgs): score + direct multi-GPU exchange
const void* kx_collect_kernel();           // k_x_collect(K3xArgs): the exchange's wait alone (host-paced re-poll)
const void* gate_kernel();                 // k_gate(GateArgs): bounded wait for the host's release word (pre-armed chain)
const void* kx_probe_kernel();             // k_x_probe(XProbeArgs): functional probe of the peer mappings
const void* k3_cluster_kernel();           // k3_cluster_score(K3cArgs)
const void* fetch_small_kernel();          // k_fetch_small(FetchArgs)
const void* k4_kernel(int nf);             // k4_viterbi<NF>(K4Args), NF in {1, 2, 4, 8}

// LDS behind the node-vector buffers of the walk: the walk's scratch (counts, column limits, step list) and, once
// the walk is over, the epilogue's scratch (candidate lists + per-family maxima) share it; the launcher sizes it for
// both.  Layout: [step list, matrix offsets, error flags: loaded once per workgroup] then, per family tile, the union
// of [counts, column limits] (walk) and [candidate lists, per-family maxima] (epilogue).
__host__ __device__ inline size_t k2_scratch_bytes(int nf, int n_leaves, int n_ops)
{
    const size_t fixed = (size_t)n_ops * (12 + 2 + 2) * 4;
    const size_t walk = (size_t)nf * n_leaves * 4 + (size_t)nf * 4 + 8;   // + the park-slot word
    const size_t epi = 8 * 64 * 4 + (size_t)nf * 8 + 16;
    return fixed + (walk > epi ? walk : epi);
}

}  // namespace cafehip
