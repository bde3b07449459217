// device_types.hpp -- parameter blocks shared by the kernel translation units (k1_matrices.hip, k2_walk16.hip,
// k2_walk4.hip, k2c_tables.hip, k_misc.hip) and the context / C ABI (cafehip.hip).  Every kernel takes ONE struct by
// value, so the context launches any of them through hipLaunchKernel with the function address a translation unit
// hands out (kernels.hpp) -- the kernels themselves stay private to their unit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/cafehip.h"
#include "schedule.hpp"

namespace cafehip {

constexpr int kMaxNodesCap = 4095;   // sanity bound on the tree (2,048 taxa); nothing is sized by it
constexpr int kMaxPrior = 1000;      // FAMILYSIZEMAX, libtree/family.h:8
constexpr int kParamRing = 8;
constexpr int kMaxSets = CAFEHIP_MAX_SETS;   // parameter sets evaluated in one pass (cafehip_eval_posterior_multi)

// one transition matrix of an evaluation: (int branch length, lambda, mu) reduced to the scalars K1 needs
struct KeyParam {
    double log_alpha, log_beta, log_coeff, coeff;
    double l2a, l2b, rho_m;   // product form (host_math.hpp KeyScalars)
    int rho_e;
    int fast_ok;
    int mode;  // host_math.hpp KeyScalars
    int bl;
    int slot;  // where K1 stores this matrix: PT[slot][KP][LD] (round 5: the device holds more than one evaluation's matrices)
};

// Per-evaluation parameter block in PINNED, device-mapped host memory (a ring of kParamRing): K1 reads its keys
// straight from it and mirrors the node -> matrix map into device memory for the later launches.  Flat, sized by
// the tree: [EvalHeader][KeyParam keys[key_cap]][int32 node_key[kMaxSets][n_nodes]][f64 prior[kMaxPrior]][f64 logprior[kMaxPrior]]
// (the prior arrays are filled, and mirrored by K1, only in an evaluation whose prior differs from the previous one's)
struct EvalHeader {
    int nkeys, n_sets, n_nodes, key_cap;
};
__host__ __device__ inline const KeyParam* eval_keys(const EvalHeader* h) { return reinterpret_cast<const KeyParam*>(h + 1); }
__host__ __device__ inline KeyParam* eval_keys(EvalHeader* h) { return reinterpret_cast<KeyParam*>(h + 1); }
__host__ __device__ inline const int32_t* eval_node_key(const EvalHeader* h, int key_cap)
{
    return reinterpret_cast<const int32_t*>(eval_keys(h) + key_cap);
}
__host__ __device__ inline int32_t* eval_node_key(EvalHeader* h, int key_cap) { return reinterpret_cast<int32_t*>(eval_keys(h) + key_cap); }
inline size_t eval_prior_offset(int key_cap, int n_nodes)
{
    const size_t o = sizeof(EvalHeader) + (size_t)key_cap * sizeof(KeyParam) + (size_t)kMaxSets * n_nodes * sizeof(int32_t);
    return (o + 7) & ~(size_t)7;
}
__host__ __device__ inline const double* eval_prior(const EvalHeader* h, size_t prior_offset)
{
    return reinterpret_cast<const double*>(reinterpret_cast<const char*>(h) + prior_offset);
}
inline size_t eval_block_bytes(int key_cap, int n_nodes) { return eval_prior_offset(key_cap, n_nodes) + 2 * (size_t)kMaxPrior * sizeof(double); }

// ---- K1 ------------------------------------------------------------------------------------------------------
struct K1Args {
    const EvalHeader* ep;        // pinned host block of this evaluation
    const double* tabA;          // ln C tables (exact form) or their exp() twins (product forms)
    const double* tabB;
    int ld_lnc;
    double* PT;
    int M, LD, KP;
    int32_t* first_zero;         // reset to INT32_MAX per set (the score kernel atomicMin's into it), or NULL
    int keys_per_block;
    int32_t* node_key_dev;       // the device's node -> matrix store [rows][n_nodes], or NULL: set s is mirrored into row set_row[s]
    int set_row[kMaxSets];
    int n_nodes, n_sets, nkeys, key_cap;
    // the prior changed: block (0,0,0) also mirrors prior[n_prior] and logprior[n_prior] (n_prior = 0 otherwise)
    int n_prior;
    size_t prior_offset;         // of the two kMaxPrior-long arrays inside the pinned block
    double* prior_dev;
    double* logprior_dev;
    int exp_variant;             // exact form: 1 / 2 = the host libm's exp restated (fused / plain build), 0 = the device library's
    // register-blocked kernel, round 6: a 1-D grid of gx * gy * gz workgroups dealt heavy tiles first, light tiles last (the
    // work of tile (s0, c0) grows with min(s, c): co-resident workgroups then add up to about the same), or balanced = 0: 3-D grid
    int gx, gy, gz, balanced;
    unsigned char tile_of_rank[128];   // tile id (by * gx + bx) of the tile ranked r by work, heaviest first (gx * gy <= 128)
};

struct FoldArgs {   // k1e_fold_error
    const double* PT;
    double* PTfold;
    const double* err;
    int err_ld, banded, dlo, dhi, C, KP, LD;
};

// ---- K2, row-per-thread kernel (k2_prune_v1) and the carrier the launchers fill -------------------------------
struct K2Args {
    const double* PT;        // [nkeys][KP][LD]  PT[k][c*LD + s] = Pr(c | s)
    const int32_t* node_key; // device [n_sets][n_nodes]
    int n_nodes;
    const double* prior;     // device [R]
    const double* logprior;  // device [R]
    const PruneOp* ops;
    int n_ops;
    const int32_t* counts;   // [Fu][n_leaves]
    int Fu;
    int n_leaves;
    int C;                   // range_max + 1 (range_min == 0)
    int R;                   // root_max - root_min + 1
    int root_min;
    int LD, KP, LDv;
    int n_slots;
    // error model (optional)
    const double* err;       // [(mfs+1)^2] row = observed
    int err_ld;
    const uint8_t* leaf_has_err;  // [n_leaves] by count column
    // per-row extents (batch mode; NULL in posterior mode)
    const int32_t* root_lo;
    const int32_t* root_hi;
    const int32_t* col_max;
    const int64_t* out_off;  // packed offsets of the root vectors
    double* out_root;
    // posterior outputs
    double* max_lik;
    int32_t* argmax;
    double* max_post;
};

// ---- K2 on the matrix cores (k2_mfma.hpp) ----------------------------------------------------------------------
struct K2MfmaArgs {
    const double* PT;
    const int32_t* node_key;   // device [n_sets][n_nodes]
    int n_nodes;
    const double* prior;       // device [R]
    const double* logprior;
    const MfmaOp* ops;
    int n_ops;
    int32_t* park_flags;   // [n_park_slots] 0 = free: a workgroup that parks in global memory owns one slot of the
    int n_park_slots;      // scratch while it runs (slots ~ 2x the resident workgroups, not one per family tile)
    int n_sets;            // gridDim.y: parameter sets evaluated in this pass; set s reads node_key[s], writes outputs at s * Fu
    int lds_parks;         // park slots [0, lds_parks) live in LDS behind the node buffer (no global round trip)
    const int32_t* counts;
    int Fu;
    int n_leaves;
    int C, R, root_min;
    int LD, KP, LDv;
    int ksteps;            // ceil(C / 4)
    int Wf, Wr;            // wave grid
    int NF;                // families per workgroup = 16 * Wf * NFT_W
    double* park;          // [n_park_slots][n_parks][NF][LDv]
    int n_parks;
    // error model
    const double* err;
    int err_ld;
    const uint8_t* leaf_has_err;
    int err_banded;        // 1: errormatrix[obs][true] is zero unless err_dlo <= true - obs <= err_dhi
    int err_dlo, err_dhi;
    const double* PTfold;  // posterior mode: error model folded into the leaf matrices (k1e_fold_error), or NULL
    // batch mode (per-row extents)
    const int32_t* root_lo;
    const int32_t* root_hi;
    const int32_t* col_max;
    const int64_t* out_off;
    double* out_root;
    int trim;              // batch mode: a tile's products stop at its largest column limit, its root step at its root sizes
    int32_t* gen_done;     // batch mode, lock-step generations: workgroups finished so far (device counter), or NULL
    int gen_size, gen_slack;   // workgroup b starts once (b / gen_size) * gen_size - gen_slack workgroups have finished
    // posterior outputs
    double* max_lik;
    int32_t* argmax;
    double* max_post;
    // compressed subtrees (schedule.hpp, CTile): factor tables [set][node table][state][LD], rows gathered like
    // matrix columns by a child of kind 2; table_off[node] = element offset of the node's table
    const double* tables;
    const int32_t* table_off;
    size_t table_set_stride;
    // debug builds (-DCAFE_K2_STAMPS): s_memtime stamps [workgroup][wave][K2_STAMP_SLOTS], else NULL and unused
    unsigned long long* stamps;
    int skip_epilogue;     // ablation (option k2_skip_epilogue): the walk ends behind its root step, the posterior outputs are NOT written
};

// ---- k2c_nodes: factor tables of compressed subtrees -----------------------------------------------------------
struct K2cArgs {
    const double* PT;
    const double* PTfold;             // or NULL
    const int32_t* node_key;          // device [n_sets][n_nodes]
    int n_nodes;
    const CTile* tiles;               // this level's tiles
    const int32_t* leaf_has_err32;    // by count-table column (leaves), one word each, or NULL
    double* tables;
    size_t table_set_stride;
    int C, LD, KP, LDv, ksteps;
    int block_threads;                // = blockDim.x (read from here: the implicit argument would be one more dependent load)
    int xcd_remap;                    // k2c_gemm: XCD x takes a contiguous eighth of the level's tiles
    // debug builds (-DCAFE_K2_STAMPS): s_memtime stamps [tile][wave (16)][8], else NULL and unused
    unsigned long long* stamps;
};

// ---- K3 --------------------------------------------------------------------------------------------------------
// Results of the synchronous path go straight to pinned, device-visible host memory (no copy kernels, no
// interrupt-driven wait): every block stores its chunk sum there, the last block to arrive (device counter)
// publishes the first-zero index and a sequence number the host spins on.
struct HostResult {
    volatile int32_t done_seq;
    int32_t first_zero[kMaxSets];
    int32_t pad;
    double chunk_sums[1];  // [n_sets][n_chunks]
};

struct K3Args {
    const double* max_post_u;
    const double* max_lik_u;
    const int32_t* fam2u;
    int F, Fu;
    double* chunk_sums;
    int32_t* first_zero;
    HostResult* host;
    int32_t* arrive;
    int32_t seq;
};

// k3_score<true> and k1_build_matrices_rb in ONE launch (k1_matrices.hip): blocks [0, k3_blocks) score, the rest build the
// (gx, gy, gz) grid of matrix tiles of the next candidates
struct K3K1Args {
    K3Args k3;
    K1Args k1;
    int k3_blocks, gx, gy;
};

// Score kernel of a SHARDED evaluation with the direct exchange (comm.hpp): every block stores its chunk sum into
// EVERY rank's exchange buffer (row of this rank), the last block adds the first-zero index, raises this rank's flag
// in every buffer, waits for the other ranks' flags in its own and hands all rows to the host.
constexpr int kXMaxWorld = 16;
struct K3xArgs {
    const double* max_post_u;
    const double* max_lik_u;
    const int32_t* fam2u;
    int F, Fu;
    int32_t* first_zero;        // device word, reset by K1
    HostResult* host;           // rows land in host->chunk_sums[world][slots + 1]
    int32_t* arrive;
    int32_t seq;                // host-visible sequence number of this evaluation
    int rank, world, slots;     // row of rank r: [slots chunk sums][first-zero index as int64 bits]
    unsigned long long xseq;    // exchange sequence number (flags), never reused
    double* rows[kXMaxWorld];               // rank r's buffer, this parity: [world][slots + 1]
    unsigned long long* flags[kXMaxWorld];  // rank r's flags, this parity: [kXMaxWorld]
    long long timeout_ticks;    // wall_clock64() ticks to wait for the peers before giving up (host sees -seq)
};

// Gate of a pre-armed launch chain (k_gate, one wave): waits until the host's word equals `want`, at most `timeout_ticks` of
// the 100 MHz clock, then writes what happened to the host's outcome word: (want << 2) | 1 released, | 2 expired.
struct GateArgs {
    const volatile unsigned long long* flag;   // pinned host memory, written by the host
    volatile unsigned long long* outcome;      // pinned host memory, written by the gate
    unsigned long long want;
    long long timeout_ticks;
};

struct XProbeArgs {   // k_x_probe
    unsigned long long* probe[kXMaxWorld];   // rank r's probe words: [kXMaxWorld]
    int rank, world, mute;
    unsigned long long nonce;
    long long timeout_ticks;
    int32_t* seen;                           // device word: peers whose store arrived
};

struct ClusterWeights {
    double w[kMaxSets];
};

struct K3cArgs {   // k3_cluster_score
    const double* max_post_u;
    const int32_t* fam2u;
    int F, Fu, K;
    ClusterWeights cw;
    double* chunk_sums;
    double* memb_sums;      // [K][n_chunks]
    int32_t* first_zero;
    double* map_out;        // [F] or NULL
    double* pz_out;         // [F][K] or NULL
};

struct FetchArgs {   // k_fetch_small
    const uint64_t* src;
    uint64_t* host_dst;
    size_t n_words;
    volatile int32_t* host_seq;
    int32_t seq;
};

// ---- K4 --------------------------------------------------------------------------------------------------------
struct K4Args {
    const double* PT;
    const int32_t* node_key;   // device [n_nodes] (set 0)
    const PruneOp* ops;
    int n_ops;
    const int32_t* counts;
    int B, n_leaves, n_nodes;
    int C, R, root_min;
    int LD, KP, LDv;
    int n_slots;
    int root;
    const int32_t* parent;     // [n_nodes]
    const int32_t* prefix;     // [n_nodes] prefix order
    const int32_t* vit_slot;   // [n_nodes] table index of internal non-root nodes, -1 otherwise
    int n_tables;
    const int32_t* root_lo;
    const int32_t* root_hi;
    const int32_t* col_max;
    int32_t* node_sizes;       // [B][n_nodes]
    unsigned short* vit_global;  // argmax tables in global scratch [grid][n_tables][NF][LDv], or NULL: in LDS
};

constexpr int K2_STAMP_SLOTS = 512;

}  // namespace cafehip
