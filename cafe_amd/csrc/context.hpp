// context.hpp -- the engine's context (everything a cafehip_ctx owns), the error / launch helpers and the debug allocator,
// shared by the translation units that implement the C ABI: cafehip.hip (set-up, launches, single-GPU entry points) and
// cafehip_comm.hip (the multi-GPU entry points).  Round-4 split of cafehip.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <unordered_map>
#include <mutex>
#include <map>
#include <array>
#include <vector>

#include "../../include/cafehip.h"
#include "host_math.hpp"
#include "comm.hpp"
#include "kernels.hpp"

using namespace cafehip;

// Debug aid (CAFEHIP_POISON=1 in the environment when the library is loaded): every device allocation is filled with
// 0xFF bytes -- NaN as a double, -1 as an int -- before anything else touches it and sits between two 64 KiB guard
// zones of the same bytes, so that a read of memory the library never wrote, or a little outside a buffer, shows up
// in the outputs instead of depending on what the allocator handed back.
static const bool g_poison = getenv("CAFEHIP_POISON") != nullptr;
constexpr size_t kPoisonGuard = 64 * 1024;
template <class T>
static hipError_t poison_malloc(T** p, size_t bytes)
{
    if (!g_poison) return hipMalloc(reinterpret_cast<void**>(p), bytes);
    char* raw = nullptr;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&raw), bytes + 2 * kPoisonGuard);
    if (e != hipSuccess) return e;
    (void)hipMemset(raw, 0xFF, bytes + 2 * kPoisonGuard);
    (void)hipDeviceSynchronize();
    *p = reinterpret_cast<T*>(raw + kPoisonGuard);
    return hipSuccess;
}
template <class T>
static hipError_t poison_free(T* p)
{
    if (!g_poison || !p) return hipFree(const_cast<void*>(static_cast<const volatile void*>(p)));
    return hipFree(reinterpret_cast<char*>(const_cast<void*>(static_cast<const volatile void*>(p))) - kPoisonGuard);
}
#define hipMalloc(p, n) poison_malloc(p, n)
#define hipFree(p) poison_free(p)

namespace cafehip_impl {

inline thread_local std::string g_err;   // cafehip_last_error(): one per thread, shared by the library's translation units

inline int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                        \
    } while (0)

// one argument struct per kernel (device_types.hpp): every launch is hipLaunchKernel(address, ..., &args)
template <class Args>
inline int launch_kernel(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const Args& args)
{
    if (!fn) return fail("internal: kernel shape not built");
    void* argv[1] = {const_cast<Args*>(&args)};
    HIP_TRY(hipLaunchKernel(fn, grid, block, argv, lds, stream));
    return 0;
}

}  // namespace cafehip_impl
using namespace cafehip_impl;

// ====================================================================================
// context
// ====================================================================================
// wave grid of a K2 launch (16x16x4 shape: nft_w family tiles per wave; 4x4x4 shape: nft_w carries G)
struct K2Cfg {
    int nft_w, nrt_w, wf, wr;
};
struct K2Cand {
    double cost;
    bool use4;
    K2Cfg cfg;
};

struct cafehip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int lds_limit = 64 * 1024;
    int n_cu = 0;

    // tree
    int n_nodes = 0, root = -1;
    std::vector<int> parent, left, right, bl_int;
    std::vector<double> bl;
    cafehip::Schedule sched;
    cafehip::PruneOp* d_ops = nullptr;
    cafehip::MfmaSchedule msched;
    cafehip::MfmaOp* d_mops = nullptr;
    double* d_park = nullptr;
    size_t park_cap = 0;
    int32_t *d_parent = nullptr, *d_prefix = nullptr, *d_vit_slot = nullptr;
    int n_vit_tables = 0;
    int k2_cfg[4] = {0, 0, 0, 0};  // NFT_W, NRT_W, Wf, Wr of the last MFMA launch
    bool k2_used_mfma = false;
    bool k2_shape4 = false;

    // subtree-state compression of the objective path (schedule.hpp, CTile; rebuilt by set_tree / set_families)
    struct CompressPlan {
        bool valid = false;
        cafehip::MfmaSchedule sched;        // walk of the reduced tree (compressed subtrees are leaves)
        cafehip::MfmaOp* d_ops = nullptr;
        int n_cols = 0;                     // index columns of the walk: surviving leaves + compressed subtree roots
        std::vector<int> col_leaf;          // per column: count-table column of the leaf, or -1 (compressed subtree)
        int32_t* d_counts = nullptr;        // [Fu][n_cols]
        uint8_t* d_col_has_err = nullptr;   // [n_cols]
        int n_nodes = 0;                    // compressed nodes
        std::vector<cafehip::CTile> tiles;
        std::vector<int> level_first;       // tiles of level l: [level_first[l], level_first[l + 1])
        std::vector<int> level_nft;         // ... of 16 * level_nft[l] states each
        std::vector<int> level_nrt;         // ... built by k2c_gemm with this many row tiles per wave (1 | 2), 0: by k2c_nodes
        cafehip::CTile* d_tiles = nullptr;
        int32_t* d_table_off = nullptr;     // [n_nodes]
        size_t table_elems = 0;             // per parameter set
        double* d_tables = nullptr;
        size_t tables_cap = 0;              // elements
        long states = 0;                    // sum of D over the compressed nodes
        long top_states = 0;                // ... over the MAXIMAL compressed nodes (the tables the walk gathers from)
    } cp;
    // run-time switches (cafehip_set_option; CAFEHIP_<NAME> in the environment is read ONCE, by cafehip_create)
    struct Options {
        int compress = 1;             // subtree-state compression of the objective path
        double compress_theta = -1;   // < 0: by table size and matrix side (rebuild_compression)
        int compress_min = 64;        // unique rows below which a table is left alone
        int compress_drop_top = 1;    // launch-bound tables: top levels whose nodes are cheaper as walk steps go back to the walk
        int compress_max_level = 0;   // > 0: nodes above this level of the compressed forest stay in the walk (sweeps)
        int errfold = 1;              // error model folded into the matrices (posterior mode)
        int errband = 1;              // banded error models as short sums of gathers
        int k1 = 0;                   // 0 auto, 1 exact form, 2 per-term product form
        int k1_kpb = 1;               // keys per K1 workgroup
        int k2 = 0;                   // 0 matrix cores, 1 row-per-thread kernel (k2_prune_v1), 2 the same in the reference's arithmetic
        int mfma = 0;                 // 0 either shape, 4 / 16: only that one
        bool have_cfg16 = false, have_cfg4 = false;
        int cfg16[4] = {0, 0, 0, 0}, cfg4[4] = {0, 0, 0, 0};   // pinned wave grids "nftw|G,nrtw,wf,wr"
        int k2tune = 1;               // measured choice of the wave grid
        int k2tune_log = 0;
        int k2slots = 1;              // park scratch by resident workgroup (0: one region per family tile)
        int ldspark = -1;             // park buffers kept in LDS (< 0: by residency)
        int vitlds = 0;               // Viterbi argmax tables in LDS
        int k2c_batch = 1;            // k2c_nodes: the child columns of a state gathered in one batch (round 3)
        int k2c_pair = -1;            // k2c_nodes: two row tiles per wave read with one 16-byte load per k-step: -1 by level size, 0 never, 1 always (round 5)
        int k2_skip_epilogue = 0;     // ABLATION: the walk without its posterior epilogue (outputs invalid; profiles/r06)
        int k1_balance = 1;           // register-blocked K1: heavy tiles first, light tiles last (round 6)
        int k2_objective_kernels = 1; // objective evaluations run the walk instantiations without the batch mode's code (k2_walk16o / k2_walk4o.hip)
        int k2_small_r = 1;           // 4-family walk: lane-per-family posterior epilogue where R <= 64 (k2_walk4s.hip)
        int k2c_gemm = -1;            // factor tables by k2c_gemm (round 6): -1 by level size, 0 never (k2c_nodes), 1 always
        int k2c_nst = 0;              // ... state tiles of 16 per workgroup: 0 by level size, else 1 | 2 | 4
        int k2c_xcd = 1;              // ... XCD x takes a contiguous eighth of a level's tiles
        int k2c_pair_min = 2;         // ... by level size: at least this many tiles per CU (crossover between 427 and 586 tiles on 256 CUs)
        int batch_trim = 1;           // batch mode: a tile's products stop at its largest column limit (round 3)
        int batch_lockstep = 1;       // batch mode: workgroups start generation by generation (L2 reuse of the edge matrices)
        int walk_lockstep = 0;        // the same pacing for the family walk of an objective evaluation
        int batch_lockstep_slack = 0; // ... a generation starts when all but this percentage of the previous ones have finished
        int exp_like_host = 1;        // K1 exact form: exp() as this host's libm computes it, when recognised (exp_like_host.hpp)
        int test_stall_ms = 0;        // test hook: cafehip_eval_posterior sleeps this long before it looks for the score
        int prearm = 0;               // the next evaluation's launches queued behind a gate while the current one runs (see Armed)
        int prefetch_kpb = 0;         // ... keys per workgroup of a build on the second stream (0: as k1kpb)
        int prefetch_where = 3;       // matrices ahead of time, parked requests: 3 trailing blocks of the score kernel's launch, 0 second stream at once, 1 the context's stream (behind the score kernel), 2 second stream behind the score kernel
    } opt;
    bool walk_compressed = false;           // the MFMA launcher walks the reduced tree (set around one launch)
    bool last_compressed = false;           // ... and the last objective evaluation did
    std::vector<int32_t> h_ucounts;         // unique rows, host copy
    std::vector<uint8_t> h_leaf_has_err;    // by count-table column
    double issued_walk = 0, issued_tables = 0;   // matrix-instruction flops issued by the last evaluation's pruning

    // families
    int F = 0, Fu = 0, n_leaves = 0;
    int range_min = 0, range_max = 0, root_min = 0, root_max = 0;
    int M = -1, S = 0, C = 0, R = 0, LD = 0, KP = 0, LDv = 0;
    int32_t* d_counts = nullptr;  // unique rows
    int32_t* d_fam2u = nullptr;
    double *d_max_lik = nullptr, *d_max_post = nullptr;
    int32_t* d_argmax = nullptr;
    double* d_chunk_sums = nullptr;
    int32_t* d_first_zero = nullptr;
    int n_chunks = 0;
    int out_sets = 1;             // parameter sets the per-family / per-chunk output buffers hold
    std::vector<int32_t> fam2u;

    // tables + matrices
    cafehip::LnCTables lnc;
    double *d_lncA = nullptr, *d_lncB = nullptr;
    double *d_expA = nullptr, *d_expB = nullptr;
    bool all_keys_fast = false, k1_product_form = false;
    bool force_exact = false;   // cafehip_set_exact_matrices: the reference's per-term arithmetic for the next builds
    // hipFuncAttributeMaxDynamicSharedMemorySize already granted, per kernel instantiation: the attribute is
    // per DEVICE, so the high-water marks live in the context (several contexts of one process may sit on
    // different GPUs)
    std::unordered_map<const void*, size_t> lds_attr;
    std::map<std::tuple<const void*, int, size_t>, int> k2_occ;   // resident workgroups per CU of a K2 launch shape
    int k2_grid = 0, k2_park_slots = 0;                          // workgroups / park slots of the last MFMA K2 launch
    int32_t* d_park_flags = nullptr;                             // park-slot ownership flags (0 = free)
    int32_t* d_gen_done = nullptr;                               // batch mode: workgroups finished (lock-step generations)
    int park_flags_cap = 0;
    double* d_PT = nullptr;
    unsigned short* d_vit = nullptr;   // Viterbi argmax tables (global scratch, grow-only)
    size_t vit_cap = 0;
    double* d_PTfold = nullptr;  // error model folded into the matrices (posterior mode), same shape as d_PT
    size_t ptfold_cap = 0;
    bool fold_current = false;
    size_t pt_keys_cap = 0;      // slots of the demand region [0, pt_keys_cap); the cache entries' slots follow (mc.first_slot)

    // per-evaluation parameters (ring of pinned staging buffers)
    EvalHeader* h_params[kParamRing] = {};   // pinned, device-mapped; sized by the tree (eval_block_bytes)
    hipEvent_t h_params_ev[kParamRing] = {};
    int ring_pos = 0;
    int key_cap = 0;                          // KeyParam slots of a ring block: kMaxSets x (n_nodes - 1)
    size_t ring_bytes = 0;
    const EvalHeader* cur_params = nullptr;   // staged block the next K1 launch reads
    int cur_slot = 0, cur_sets = 1, cur_prior_n = 0;
    bool prior_on_device = false;             // d_prior / d_logprior hold prior_seen
    int32_t* d_node_key = nullptr;            // [kMaxSets][n_nodes] mirror of the staged node -> matrix map (K1 writes it)
    double *d_prior = nullptr, *d_logprior = nullptr;   // [kMaxPrior] the prior of the evaluations and its logarithms
    std::vector<int> node_key;
    std::vector<double> stage_l, stage_m;   // key dedup scratch of stage_params (kept: no allocation per evaluation)
    std::vector<int> stage_b;
    std::vector<int> stage_node_key;        // mc_stage: a candidate set's node -> slot map until the set is accepted
    std::vector<double> prior_seen, logprior_seen;   // the last prior staged and its logarithms
    int nkeys = 0;
    bool have_matrices = false;

    // ---- matrices of parameter sets that MAY be evaluated next (round 5: cafehip_prefetch_matrices) ------------------
    // The optimiser knows the handful of points it can ask for next before the score of the current one is back
    // (libcommon/fminsearch.cpp:198-237 are functions of the simplex).  Their matrices are built on a second, low-priority
    // stream while the current evaluation runs, into slots of d_PT behind the demand region, each set keyed exactly as the
    // reference keys its cache -- (int branch length, lambda, mu) per node with the doubles compared bit for bit
    // (libtree/birthdeath.h:26-31, cafe/cafe_tree.c:374-391) -- and an evaluation that finds its set there binds the nodes
    // to those matrices and launches no K1: the matrix build leaves the evaluation's serial chain.  Same kernel, same
    // per-key arithmetic as a build on demand: the matrices, and every value downstream, are bit-identical.
    struct MatrixCache {
        struct Entry {
            bool valid = false;
            std::vector<double> nl, nm;   // the set as handed over, non-root nodes compared exactly
            std::vector<int> node_key;    // node -> ABSOLUTE matrix slot (host copy; the device copy is row kMaxSets + e of d_node_key)
            int nkeys = 0;
            bool folded = false;          // d_PTfold holds the error-folded twins of its matrices
            bool ready_known = false;     // its build has been seen complete (no stream wait needed any more)
            hipEvent_t ready = nullptr;   // recorded on the speculation stream behind its build
            unsigned long tick = 0;       // last use (least recently used entry is replaced)
        };
        int want_entries = 12;            // option matrix_cache=<entries> (0: off)
        size_t max_bytes = (size_t)1 << 30;
        bool broken = false;              // the entries did not fit max_bytes / an allocation failed: prefetches are ignored
        std::vector<Entry> e;             // empty until the first prefetch
        int kpe = 0;                      // slots per entry: the matrices one set can need (n_nodes - 1)
        size_t first_slot = 0;            // of entry 0 in d_PT / d_PTfold
        size_t slots_allocated = 0;       // cache slots the current d_PT allocation holds
        int bound = -1;                   // entry the pruning launches read now; -1: the demand region
        unsigned long tick = 0;
        hipStream_t stream = nullptr;
        hipEvent_t chain_end = nullptr;   // prefetch_where=2: recorded on the context's stream behind the score kernel
        int pending_sets = 0;             // request parked until the next evaluation's launches have gone out
        std::vector<double> pending_l, pending_m;
        long requested = 0, built = 0, hits = 0, misses = 0, evicted = 0, waited = 0, launches = 0;
    } mc;
    // ---- pre-armed chain (round 5, option prearm): the NEXT evaluation's launches queued behind a gate kernel while the
    // current one runs; the next cafehip_eval_posterior stages its parameters into the block the chain reads and releases
    // the gate with one store.  Only the plain synchronous single-set path, only once the wave grid is settled, only while
    // nobody announces parameter sets (a search that does is served from the matrix store instead).
    struct Armed {
        bool on = false;
        int slot = 0;                     // ring block its K1 reads
        int nkeys = 0;
        bool all_fast = false;
        int32_t seq = 0;                  // sequence number its score kernel publishes
        unsigned long long gate = 0;      // value that releases it
    } armed;
    unsigned long long* h_gate = nullptr;     // pinned: [0] release word, [16] outcome word (their own cache lines)
    unsigned long long gate_seq = 0;
    unsigned long long released_gate = 0;     // gate of the chain the current evaluation rode on (0: launched normally)
    long mc_requested_seen = 0;               // mc.requested at the previous evaluation
    long prearm_used = 0, prearm_wasted = 0, prearm_expired = 0;
    const int32_t* cur_node_key = nullptr;   // the node -> matrix map the pruning launches read (d_node_key, or a cache entry's row of it)
    int node_key_rows = 0;                   // rows of d_node_key: kMaxSets demand rows + one per cache entry
    bool fz_clean = false;                   // d_first_zero was left at INT32_MAX by the last score kernel (k3_score<true> / k3_score_x)

    // error model
    double* d_err = nullptr;
    int err_mfs = -1;
    int err_banded = 0, err_dlo = 0, err_dhi = 0, err_band_width = 0;
    uint8_t* d_leaf_has_err = nullptr;
    int32_t* d_leaf_has_err32 = nullptr;   // the same flags as 32-bit words: k2c_nodes reads them with scalar loads

    // pinned, device-visible result block of the synchronous path
    HostResult* h_result = nullptr;
    size_t h_result_chunks = 0;
    uint64_t* h_fetch = nullptr;   // cafehip_fetch_small: [0] sequence word, [1..] data
    size_t h_fetch_words = 0;
    int32_t fetch_seq = 0;
    int32_t* d_arrive = nullptr;
    int32_t host_seq = 0;          // last sequence number handed to a score kernel
    int32_t eval_seq = 0;          // ... the one the CURRENT evaluation's score kernel publishes (a pre-armed chain behind it holds a later one)

    // timing
    bool timing = false, timing_pending = false;
    hipEvent_t ev[4] = {};
    double last_ms[3] = {0, 0, 0};
    double last_tables_ms = 0;          // part of last_ms[1]: the k2c_nodes launches (compressed subtrees)
    hipEvent_t ev_mid = nullptr;
    bool ev_mid_used = false;
    double last_batch_ms = 0;   // pruning launch of the last cafehip_eval_root_likelihoods call
    int k2_nf = 0, k2_block = 0;
    size_t k2_lds = 0;
    // measured choice of the K2 wave grid (posterior path): see launch_k2_mfma
    struct {
        int n_items = -1;
        std::vector<K2Cand> cands;
        std::vector<float> best_ms;
        int cur = 0, round = 0, locked = -1;
        int reps_launched = 1;   // launches inside the pending measurement
        bool pending = false;
        hipEvent_t e0 = nullptr, e1 = nullptr;
    } tune;
    std::string desc;

    // multi-GPU: one process per GPU of a node (comm.hpp).  Set by cafehip_comm_init / cafehip_comm_set_blocks.
    CommLink* link = nullptr;
    int comm_mode = 0;                          // option "comm": 0 auto (direct when every rank mapped every buffer), 1 rccl, 2 direct
    std::vector<int32_t> blk_lo, blk_hi;        // every rank's block [lo, hi) of the global table
    int x_slots = 0;                            // chunk slots of a rank's packed row
    unsigned long long x_seq = 0;               // exchange sequence number (direct mode); re-aligned to 0 by every collective
                                                // cafehip_comm_set_blocks / cafehip_comm_resync, advanced only by a launch that went out
    K3xArgs x_last;                             // the last direct exchange's arguments: what a host-paced re-poll waits on
    int comm_agreed_mode = 0;                   // what the ranks agreed on in cafehip_comm_init: 2 direct, 1 rccl
    int comm_injected = 0;                      // CAFEHIP_COMM_INJECT made this rank mute in the probe (tests)
    long x_repolls = 0;                         // k_x_collect launches (a peer was more than a wait slice late)
    double *d_packed = nullptr, *d_gathered = nullptr;   // RCCL mode: [slots + 1] and [world][slots + 1]
    int packed_slots = 0;
    hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;
    bool ev_x_pending = false;
    double last_exchange_ms = 0;                // RCCL mode with timing on: all-gather + result pick-up, events on the stream
    double x_host_seconds = 0;                  // host time inside the exchange step (RCCL mode: launch + pick-up)
    long x_calls = 0;
    int x_mode_used = 0;                        // exchange mode of the last sharded evaluation (1 rccl, 2 direct)
    double setup_ms[4] = {0, 0, 0, 0};          // last cafehip_set_families: row dedup, compression plan, uploads + allocation, total
#ifdef CAFE_K2_STAMPS
    unsigned long long* d_stamps = nullptr;   // debug timeline of the last K2 launch (tools/k2_stamps.py)
    size_t stamps_cap = 0;
#endif
};

// ---- shared between the translation units of the context (cafehip.hip: set-up, launches, single-GPU entry points;
//      cafehip_comm.hip: the multi-GPU entry points) ---------------------------------------------------------------
namespace cafehip_impl {
// one in-kernel wait of the direct exchange (the host repeats it until comm_timeout_s is over)
inline double x_wait_slice_s() { return std::min(1.0, comm_timeout_s()); }
// K1 -> (error fold) -> table levels -> walk -> score kernel of ONE evaluation on the context's stream (cafehip.hip)
int eval_device(cafehip_ctx* c, const double* node_lambda, const double* node_mu, const double* prior, double* d_chunk_sums,
                int32_t* d_first_zero, bool host_out = false, int n_sets = 1, bool direct_exchange = false);
// elapsed times of the last evaluation's launches (blocks until its last event has completed)
int collect_kernel_ms(cafehip_ctx* c);
// a pre-armed chain is let go (it repeats the previous evaluation; its result is ignored): before anything else uses the stream
void disarm(cafehip_ctx* c);
// the current evaluation rode on a pre-armed chain whose gate had expired under the host's hands: its result is void
bool armed_chain_expired(cafehip_ctx* c);
}  // namespace cafehip_impl
