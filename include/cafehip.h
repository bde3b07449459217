/*
 * cafehip.h -- C ABI of the MI355X (gfx950) likelihood engine for CAFE's
 * per-family birth-death hot path.
 *
 * The reference (hahnlab/CAFE v4.2.1) has no FFI; the seam a maintainer would cut
 * is the pair of plain functions its optimiser objectives call once per
 * evaluation:
 *
 *   void   reset_birthdeath_cache(pCafeTree, int k, family_size_range*)   cafe/cafe.h:90, cafe/cafe_main.c:319-326
 *   double get_posterior(pCafeFamily, pCafeTree, std::vector<double>&)    cafe/lambda.h:75, cafe/lambda.cpp:691-724
 *
 * plus, for the Monte-Carlo null / report phase,
 *
 *   matrix cafe_conditional_distribution(pCafeTree, family_size_range*, int, int)  cafe/conditional_distribution.h:15
 *   void   cafe_tree_p_values(pCafeTree, std::vector<double>&, matrix&, int)       cafe/pvalue.h:16
 *
 * Every entry point below names the reference interface it replaces.
 * Conventions: plain pointers and sizes only; all host arrays are caller-owned;
 * the context owns all device memory; no exception crosses the ABI; functions
 * returning int give 0 on success and <0 on error with the message available
 * from cafehip_last_error().  One context drives ONE GPU (one process per GPU;
 * families are sharded above this layer).  A context is not
 * re-entrant.  There is no CPU fallback: without a usable HIP device
 * cafehip_create fails.
 */
#ifndef CAFEHIP_H
#define CAFEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cafehip_ctx cafehip_ctx;

/* Parameter sets one pass over the table can evaluate (cafehip_eval_posterior_multi). */
#define CAFEHIP_MAX_SETS 8

/* Families per partial-sum chunk of the score reduction (fixed so that the
 * summation order does not depend on how families are sharded over GPUs). */
#define CAFEHIP_CHUNK 256

/* ABI version of this header (bumped on any signature change). */
int cafehip_abi_version(void);

/* Create a context on HIP device `device_id` (hipSetDevice ordinal).
 * Replaces: the process-global state the reference keeps in `probability_cache`
 * (cafe/cafe_main.c:16) and `cache` (libtree/birthdeath.c:32). */
int cafehip_create(cafehip_ctx **out, int device_id);
void cafehip_destroy(cafehip_ctx *ctx);

/* Run-time switches of a context (A/B runs, tuning sweeps, tests of the alternative kernels).  Every one has a
 * default that is the product path; none changes a value beyond what DESIGN.md states for it.  key / value:
 *   compress        0|1        subtree-state compression of the objective path (1)
 *   compress_theta  0..1       share of the unique rows a node's distinct states may reach (by table size and matrix width;
 *                              1.0 = every node below the root is a factor table: the default on matrices of >= 200 columns)
 *   compress_min    n          unique rows below which a table is left alone (64)
 *   compress_drop_top 0|1      launch-bound tables (one round of walk workgroups): top levels of the compressed forest whose
 *                              nodes are cheaper as walk steps than the level's launch go back to the walk (1)
 *   compress_max_level n       nodes above level n of the compressed forest stay in the walk (0: no limit; sweeps)
 *   errfold         0|1        error model folded into the matrices in an objective evaluation (1)
 *   errband         0|1        banded error models as short sums of column gathers (1)
 *   k1              auto|exact|perterm   arithmetic form of the matrix build (auto: register-blocked product form)
 *   k1kpb           n          matrices per K1 workgroup (1)
 *   k1_balance      0|1|2|3|11 order in which the register-blocked build deals its (tile, key) pairs: 1 = heaviest first where a
 *                              launch is at least three workgroups per CU, grid order otherwise (default); 0 grid order; forced:
 *                              2 alternating, 3 heaviest first, 11 heavy half / light half.  Same matrices bit for bit
 *   k2              auto|v1|v1ref   pruning kernel: matrix cores | row-per-thread vector FMA | row-per-thread in the
 *                              reference's arithmetic (separate multiply and add per term: with k1=exact the oracle's bits)
 *   mfma            auto|4|16  matrix instruction shape of the walk
 *   k2cfg, k2cfg4   "a,b,wf,wr"  pin the wave grid of the 16x16x4 / 4x4x4 kernel (empty: measured choice)
 *   k2tune          0|1        measure the wave grids on the first evaluations of a table (1)
 *   k2tune_log      0|1        print the measured table to stderr
 *   k2slots         0|1        park scratch owned per resident workgroup (1) | one region per family tile
 *   ldspark         n          park buffers kept in LDS (empty: by residency)
 *   vitlds          0|1        Viterbi argmax tables in LDS (0: global scratch)
 *   k2c_gemm        -1|0|1     factor tables by k2c_gemm, the chunk-pipelined table kernel of round 6 (default), 0 = k2c_nodes
 *   k2c_nst         0|1|2|4    k2c_gemm: state tiles of 16 per workgroup (0: by level size and matrix width)
 *   k2c_xcd         0|1        k2c_gemm: on levels of at least four tiles per CU, XCD x takes a contiguous eighth of the tiles (1)
 *   k2_skip_epilogue 0|1       ABLATION: the walk ends behind its root step, the posterior outputs are NOT written (0)
 *   k2c_batch       0|1        k2c_nodes gathers the child columns of a state in one batch (1)
 *   k2c_pair        -1|0|1     factor-table kernel deals a wave TWO row tiles and reads both with one 16-byte load per k-step
 *                              (half the vector-memory instructions, four tiles resident per CU instead of two): -1 = on levels
 *                              of at least k2c_pair_min tiles per CU (default), 0 = never, 1 = always.  Bit-identical either way
 *   k2c_pair_min    n          ... tiles per CU from which a level uses pairs (2)
 *   batch_trim      0|1        batch mode: a tile's products stop at its largest column limit (1)
 *   batch_lockstep  0|1        batch mode: workgroups start generation by generation, a generation = as many as the
 *                              chip holds at once, so that co-resident tiles stream the same edge matrix through L2 (1)
 *   batch_lockstep_slack 0..100  ... a generation starts when all but this percentage of the earlier ones finished (0)
 *   walk_lockstep   0|1        the same pacing for the family walk of an objective evaluation (0: measured slower)
 *   exp_like_host   0|1        exact-form matrices call exp() as THIS HOST's libm computes it, restated for the device, when
 *                              one of its two builds matches std::exp at first use (1); 0: the device library's exp
 *   prearm          0|1        the next evaluation's launches queued behind a gate while the current one runs
 *                              (cafehip_prearm_stats; 0)
 *   prefetch_where  0..3       where the builds of announced parameter sets run: 3 trailing blocks of the score kernel's
 *                              launch (default), 0 second stream at once, 1 the context's stream behind the score kernel,
 *                              2 second stream behind an event (A/B runs: profiles/r05/matrices_ahead_of_time_ab.txt)
 *   prefetch_kpb    n          matrices per workgroup of a build on the second stream (0: as k1kpb)
 *   test_stall_ms   n          test hook: cafehip_eval_posterior sleeps n ms before it looks for the score (0)
 *   matrix_cache    n          entries of the store of matrices built ahead of time (cafehip_prefetch_matrices; 12, 0: off)
 *   matrix_cache_mb n          ... and its size limit in MiB (1024)
 *   comm            auto|direct|rccl   exchange mode of sharded evaluations (multi-GPU section below)
 *   k2_objective_kernels 0|1   objective evaluations run the walk instantiations compiled without the batch mode's and the
 *                              unfolded error model's code (k2_walk16o / k2_walk4o.hip; 1)
 *   k2_small_r      0|1        4-family walk of a table with at most 64 root sizes: lane-per-family posterior epilogue
 *                              (k2_walk4s.hip; 1)
 * The same names, upper-cased behind CAFEHIP_ (CAFEHIP_COMPRESS=0 ...), are read from the environment ONCE, by
 * cafehip_create; nothing reads the environment during an evaluation.  Options that change the compression plan
 * rebuild it.  No reference counterpart. */
int cafehip_set_option(cafehip_ctx *ctx, const char *key, const char *value);
/* The current value of a switch that a caller may want to put back after changing it for one call (k2, k1, compress,
 * errfold, matrix_cache, prefetch_where, comm), as the string cafehip_set_option takes. */
int cafehip_get_option(cafehip_ctx *ctx, const char *key, char *value, size_t value_bytes);

/* Run all subsequent work of this context on the caller's HIP stream (a hipStream_t passed as
 * void*).  NULL is the HIP legacy default stream -- the handle torch reports for its default
 * stream -- NOT "no stream": until this is called the context uses a private non-blocking stream. */
int cafehip_set_stream(cafehip_ctx *ctx, void *hip_stream);
/* The stream the context currently runs on (a hipStream_t as void*), for a caller that wants to enqueue its own
 * work -- e.g. the collective of the multi-GPU exchange -- behind the context's kernels. */
int cafehip_get_stream(cafehip_ctx *ctx, void **hip_stream);

/* Tree topology in the reference's nlist numbering (in-order: even ids are
 * leaves, odd ids internal; cafe/cafe_commands.cpp:2028-2051).  parent[root] = -1,
 * left/right = -1 for leaves.  Up to 4,095 nodes (2,048 taxa; a sanity bound -- every buffer is sized by the
 * tree).  Branch lengths are truncated to int inside, as the reference's cache key does (libtree/birthdeath.h:26-31, cafe/cafe_tree.c:376).
 * Replaces: cafe_tree_new + tree_build_node_list as consumed by
 * cafe_tree_set_birthdeath (cafe/cafe_tree.c:461-483). */
int cafehip_set_tree(cafehip_ctx *ctx, int n_nodes, const int32_t *parent, const int32_t *left,
                     const int32_t *right, const double *branchlength);

/* Count table: F x n_leaves int32, row-major, column j <-> leaf node id 2*j.
 * ref[i] = lowest index with an identical row (cafe/cafe_family.c:9-34), or NULL
 * to have it computed (hashed, same result).  Ranges as libtree/family.h:10-15 /
 * init_family_size (cafe/cafe_family.c:357-364); range_min must be 0.
 * Replaces: the pCafeFamily argument of get_posterior + copy_range_to_tree
 * (cafe/cafe_main.c:52-60) + cafe_family_set_size (cafe/cafe_family.c:211-234). */
int cafehip_set_families(cafehip_ctx *ctx, int F, int n_leaves, const int32_t *counts,
                         const int32_t *ref, int range_min, int range_max, int root_min,
                         int root_max);

/* Host wall-clock of the last cafehip_set_families call, the one-time set-up SURVEY.md section 8(d) asks to be reported
 * apart from the evaluations: ms[0] duplicate-row detection (ref), ms[1] subtree-state compression plan (state
 * numbering + tiles + upload), ms[2] uploads and allocations (count table, outputs, ln C tables when the matrix side
 * changed), ms[3] the whole call. */
int cafehip_last_setup_ms(cafehip_ctx *ctx, double ms[4]);

/* Error model: errormatrix[(mfs+1) x (mfs+1)] row = observed, col = true
 * (libtree/family.h:31-38), leaf_has_model[n_nodes] marks leaves that carry it
 * (cafe/error_model.cpp:206-229).  Pass errormatrix = NULL to remove.
 * Replaces: the errormodel branch of initialize_leaf_likelihoods
 * (cafe/cafe_tree.c:196-203). */
int cafehip_set_error_model(cafehip_ctx *ctx, int mfs, const double *errormatrix,
                            const uint8_t *leaf_has_model);

/* One objective evaluation == reset_birthdeath_cache + get_posterior
 * (cafe/cafe_main.c:319-326, cafe/lambda.cpp:691-724).
 * node_lambda/node_mu[n_nodes] exactly as cafe_shell_set_lambdas leaves them
 * (cafe/cafe_shell.c:31-38; mu < 0 selects the lambda-only form).  prior[R] =
 * prior_rfsize[0..R), R = root_max-root_min+1.
 * Outputs: *score = sum_i log(max_posterior_i) in family order, or -inf when some
 * family has max_likelihood == 0, in which case *first_zero_family is the lowest
 * such index (mirrors the throw at cafe/lambda.cpp:715-720), else -1.
 * Optional per-family arrays (NULL to skip): max_lik[F], argmax_root[F]
 * (index into the root range, as pitem->maxlh, cafe/lambda.cpp:673-676),
 * max_post[F].
 * The first ~20-25 evaluations after a table / tree / error model is set run different launch shapes of the
 * pruning kernel while the library times them and keeps the fastest; every shape returns bit-identical values,
 * so those evaluations are ordinary ones (option k2tune=0 disables the measurement). */
int cafehip_eval_posterior(cafehip_ctx *ctx, const double *node_lambda, const double *node_mu,
                           const double *prior, double *score, int32_t *first_zero_family,
                           double *max_lik, int32_t *argmax_root, double *max_post);

/* n_sets (1..CAFEHIP_MAX_SETS) objective evaluations in ONE pass over the table: set s uses
 * node_lambda[s * n_nodes ...], node_mu[s * n_nodes ...]; all share the prior.  The matrices of all sets are built by
 * one launch (identical (branch length, lambda, mu) keys are shared across sets), the pruning launch gains a set
 * dimension, the score reduction too.  scores[s] / first_zero_family[s] are bit-identical to what
 * cafehip_eval_posterior returns for set s alone.  What it is for: a table that fills only a fraction of the chip
 * -- the four candidate vertices of a Nelder-Mead iteration (libcommon/fminsearch.cpp:198-237), the points of a
 * `lambda -r` grid (cafe/lambda.cpp:192-231), the clusters of the -k model -- cost little more than one evaluation.
 * Fails (caller falls back to single calls) when the sets need more than 256 distinct matrices or the matrix side
 * is beyond the matrix-core kernels. */
int cafehip_eval_posterior_multi(cafehip_ctx *ctx, int n_sets, const double *node_lambda, const double *node_mu,
                                 const double *prior, double *scores, int32_t *first_zero_family);
/* n objective evaluations ONE AFTER THE OTHER (set i = node_lambda/node_mu + i * n_nodes, all under `prior`): evaluation i is
 * complete -- its score on the host, in scores[i] -- before evaluation i + 1 is staged, i.e. exactly n calls of
 * cafehip_eval_posterior (or, sharded != 0, of cafehip_eval_posterior_sharded on every rank), minus the caller's per-call
 * overhead.  What it is for: loops whose points are known up front but whose evaluations must not share a pass -- the grid of
 * `lambda -r` when a table fills the chip (cafe_lambda_distribution, cafe/lambda.cpp:192-231, evaluates its points one by
 * one), likelihood profiles.  (Measured for bench.py's timed steps in round 5: no faster than one ctypes call per step --
 * 0.1178 against 0.1154 ms at configs[1] -- so the bench keeps its loop; the entry point saves a caller the loop, not time.) */
int cafehip_eval_posterior_sequence(cafehip_ctx *ctx, int n, const double *node_lambda, const double *node_mu,
                                    const double *prior, double *scores, int32_t *first_zero_family, int sharded);
/* One evaluation of the k-cluster objective (`lambda -k`): cafe_get_clustered_posterior (cafe/cafe_main.c:165-253) over
 * cafe_tree_clustered_likelihood (cafe/cafe_tree.c:704-850).  Cluster k prunes every family with its own rates
 * (node_lambda/node_mu + k * n_nodes; K <= CAFEHIP_MAX_SETS) in the same pass as the other clusters; per family
 * MAP_k = max posterior_k * weights[k], membership p_z[k] = MAP_k / sum_k MAP_k, MAP = sum_k p_z[k] * MAP_k;
 * *score = sum over families of log MAP (-inf, with *first_zero_family the lowest index, if some MAP == 0 -- the
 * reference stops its loop there, :231-240).  membership_sums[k] = sum over families of p_z[k]: the reference's new
 * weights are membership_sums[k] / F (:243-245).  Optional per-family outputs: family_map[F], family_membership[F*K]
 * (param->MAP, param->p_z_membership). */
int cafehip_eval_clustered_posterior(cafehip_ctx *ctx, int K, const double *node_lambda, const double *node_mu,
                                     const double *weights, const double *prior, double *score,
                                     int32_t *first_zero_family, double *membership_sums, double *family_map,
                                     double *family_membership);

/* Workgroups of the last pruning launch and the device's compute units (how full a single evaluation makes the chip). */
int cafehip_launch_info(cafehip_ctx *ctx, int *k2_workgroups, int *compute_units);

/* Matrix-instruction flops issued by the pruning of the last objective evaluation (tile padding included), for
 * roofline accounting: `walk` = the family walk (one product per internal child edge of the walked tree and family
 * slot), `tables` = the factor tables of compressed subtrees (families that agree on the counts below a node share
 * its vector; the library builds the product with the node's edge matrix once per distinct state -- bit-identical
 * values, less work; option compress=0 disables).  No reference counterpart. */
int cafehip_last_issued_flops(cafehip_ctx *ctx, double *walk, double *tables);

/* Same evaluation, but nothing is copied back and nothing synchronises: the
 * per-chunk partial sums (cafehip_num_chunks doubles; chunk c covers families
 * [c*CAFEHIP_CHUNK, (c+1)*CAFEHIP_CHUNK)) and the first-zero index (INT32_MAX if
 * none) are left in caller-provided DEVICE buffers on the context's stream, ready
 * for an RCCL all-gather/all-reduce.  Used by the multi-GPU driver. */
int cafehip_eval_posterior_async(cafehip_ctx *ctx, const double *node_lambda,
                                 const double *node_mu, const double *prior,
                                 double *d_chunk_sums, int32_t *d_first_zero);
int cafehip_num_chunks(cafehip_ctx *ctx);

/* S x S row-major transition matrix bound to `node`'s edge by the last evaluation
 * (== node->birthdeath_matrix->values, libtree/birthdeath.h:8-22).  *S_out = M+1. */
int cafehip_get_matrix(cafehip_ctx *ctx, int node, double *out, int *S_out);
/* Side of the matrices of this context (M+1, M = max(range_max, root_max)). */
int cafehip_matrix_size(cafehip_ctx *ctx);

/* Matrices ahead of time (round 5).  An optimiser knows the few points it may ask for next BEFORE the score of the current
 * one is back -- the reflection, expansion and contraction points of a Nelder-Mead step are functions of the simplex
 * (libcommon/fminsearch.cpp:198-237) -- and the matrix build of an evaluation (compute_birthdeath_rates for every key of
 * cafe_tree_set_birthdeath, cafe/cafe_tree.c:461-483) depends on nothing else.  This call hands the library up to
 * CAFEHIP_MAX_SETS candidate parameter sets (set s = node_lambda/node_mu + s * n_nodes, as for cafehip_eval_posterior).
 * Their matrices are built on a second, low-priority stream into a device store keyed like the reference's cache:
 * (int branch length, lambda, mu) per node, the doubles compared exactly (libtree/birthdeath.h:26-31,
 * cafe/cafe_tree.c:380-382).  A later cafehip_eval_posterior / cafehip_eval_posterior_sharded of one of the sets binds the
 * nodes to those matrices and launches no matrix build: the evaluation's serial chain starts at the pruning.  The same
 * kernel builds them with the same per-key arithmetic, so every value of such an evaluation is bit-identical to one that
 * builds on demand; a set that was not announced, or was replaced meanwhile (least recently used of `matrix_cache`
 * entries, default 12, at most `matrix_cache_mb` MiB, default 1024), is simply built on demand.  A hint: never an error
 * to announce points that are not evaluated.
 * when = CAFEHIP_PREFETCH_NOW: launched before the call returns.
 * when = CAFEHIP_PREFETCH_BEHIND_NEXT_EVALUATION: kept until the next evaluation's own launches are in the queue, so that
 *        the host work of staging the candidates does not delay it; replaces an earlier request nobody picked up. */
#define CAFEHIP_PREFETCH_NOW 0
#define CAFEHIP_PREFETCH_BEHIND_NEXT_EVALUATION 1
int cafehip_prefetch_matrices(cafehip_ctx *ctx, int n_sets, const double *node_lambda, const double *node_mu, int when);
/* out: [0] sets announced, [1] sets built, [2] evaluations that found their matrices, [3] evaluations that looked and did
 * not, [4] entries replaced, [5] hits that still had to wait for the build, [6] build launches, [7] entries of the store */
#define CAFEHIP_MATRIX_CACHE_STATS 8
int cafehip_matrix_cache_stats(cafehip_ctx *ctx, long out[CAFEHIP_MATRIX_CACHE_STATS]);

/* Pre-armed chain (round 5, option prearm=1; OFF by default -- worth 1 us per evaluation inside a steady loop, but the loop's end
 * pays for it: whoever synchronises the stream next waits out the gate's 20 ms or a repeated evaluation).  While a synchronous single-set evaluation runs, the launches of
 * the NEXT one -- matrix build, table levels, walk, score kernel -- are queued behind a one-wave gate kernel; the next
 * cafehip_eval_posterior stages its parameters into the block that chain reads and starts it with one store to pinned memory
 * instead of a launch.  Same kernels on the same inputs: identical values.  The gate waits at most 20 ms (then the chain
 * repeats the previous evaluation and its result is ignored), every other entry point lets a waiting chain go first, and a
 * chain is armed only once the wave grid of the table is settled and while nobody announces parameter sets.
 * out = {evaluations that rode on a pre-armed chain, chains let go unused, chains whose gate expired under a release}. */
int cafehip_prearm_stats(cafehip_ctx *ctx, long out[3]);

/* Rebuild the matrices for (node_lambda, node_mu) without scoring
 * (== reset_birthdeath_cache alone, cafe/cafe_main.c:319-326). */
int cafehip_reset_birthdeath_cache(cafehip_ctx *ctx, const double *node_lambda,
                                   const double *node_mu);

/* on != 0: the following matrix builds use the exact form -- exp() per term, running product for coeff^j,
 * no contraction: the IEEE operation sequence of the reference's compute_birthdeath_rates
 * (libtree/birthdeath.c:34-73, 238-286) -- instead of the faster product form (<= 5e-12 relative apart).
 * The report phase asks for it: Monte-Carlo draws compare uniform numbers with cumulative sums of matrix rows
 * (cafe/cafe_tree.c:533-569) and viterbi_sum_probabilities compares entries with exact == and <
 * (cafe/viterbi.cpp:60-67), so its matrices should round like the reference's.  Objective evaluations may
 * keep the default. */
int cafehip_set_exact_matrices(cafehip_ctx *ctx, int on);

/* Root likelihood vectors for a batch of B count rows with per-row extents, using
 * the matrices of the last evaluation / reset: row b is scored with root rows
 * [root_lo[b], root_hi[b]] and columns [0, col_max[b]].  out is packed,
 * sum_b (root_hi[b]-root_lo[b]+1) doubles.  The error model is NOT applied
 * (the reference drops it on tree copies, cafe/cafe_tree.c:485-494).
 * Replaces: compute_tree_likelihoods + get_likelihoods as called from
 * get_random_probabilities (cafe/conditional_distribution.cpp:10-44) and
 * cafe_tree_p_values (cafe/pvalue.cpp:143-154). */
int cafehip_eval_root_likelihoods(cafehip_ctx *ctx, int B, const int32_t *counts,
                                  const int32_t *root_lo, const int32_t *root_hi,
                                  const int32_t *col_max, double *out);

/* Viterbi ancestral states (max-product pass + backtrack) for a batch of B count rows with per-row
 * extents, using the matrices of the last evaluation / reset.  node_sizes: B x n_nodes int32 out
 * (leaf entries repeat the counts, internal entries are the most likely sizes; strict '>' so the
 * first maximum wins).  Rows whose products are all zero keep index 0 (the reference leaves stale
 * memory there, cafe/viterbi.cpp:296-300).
 * Replaces: cafe_tree_viterbi (cafe/viterbi.cpp:494-516, compute :208-320, backtrack :322-351)
 * under the per-family ranges of cafe_family_set_size_with_family_forced (cafe/cafe_family.c:236-255). */
int cafehip_viterbi(cafehip_ctx *ctx, int B, const int32_t *counts, const int32_t *root_lo,
                    const int32_t *root_hi, const int32_t *col_max, int32_t *node_sizes);

/* ---- multi-GPU: one process per GPU of ONE node, families sharded, the exchange behind this ABI ---------------------
 * The reference's get_posterior is a map over families followed by a sum (cafe/lambda.cpp:698-722); sharded, every
 * rank maps its contiguous, chunk-aligned block of the table and the ranks exchange ONE packed row each -- the
 * per-chunk partial sums of the block and the index of its first zero-likelihood family -- after which every rank
 * repeats the same fixed-order sum: the score is bit-identical for any number of GPUs, so every rank's Nelder-Mead
 * takes the same decisions and nothing is broadcast.  Two exchange modes (option "comm" = auto | direct | rccl):
 *   direct  the score kernel of every rank stores its row straight into an uncached buffer of every other rank
 *           (hipIpc-mapped, over xGMI) and waits for the others' flags: no collective launch, the sharded evaluation
 *           is the same launches as the single-GPU one.  Used when every rank could map every buffer AND the
 *           functional probe of cafehip_comm_init heard every peer on every rank.
 *   rccl    ONE ncclAllGather of the rows on the context's stream (librccl resolved with dlopen when first needed),
 *           picked up with cafehip_fetch_small.
 * Rendezvous, barriers and the all-gather of host blocks (report phase) run over a POSIX shared-memory segment named
 * by the 128-byte id: rank 0 obtains one and hands it to the others by any means (file, environment, pipe).
 * Every call below except cafehip_comm_unique_id / cafehip_comm_info is collective: all ranks make it, in the same
 * order.  No reference counterpart (the reference is one process; its threads split the same loop).
 * Status: exercised with 1-3 processes sharing one GPU (bit-identical to one context) and with injected failures; not
 * yet run between two physical GPUs -- tests/test_gpu_comm.py holds the rank-per-device test for the first node. */
#define CAFEHIP_COMM_ID_BYTES 128
int cafehip_comm_unique_id(void *out_id /* CAFEHIP_COMM_ID_BYTES */);
/* Joins the ranks: rendezvous, buffers mapped, then a FUNCTIONAL probe -- every rank's one-workgroup kernel stores a
 * nonce into every peer's buffer through the mapping and waits at most 1 s for theirs -- and one collective decision
 * through the mailboxes: direct only if every rank heard every peer, else RCCL (communicator formed here, by every
 * rank), else the call fails ON EVERY RANK with the same reason.  No rank can discover an unreachable peer later,
 * inside an evaluation.  Option "comm" = rccl / direct set BEFORE this call restricts the choice. */
int cafehip_comm_init(cafehip_ctx *ctx, int rank, int world, const void *unique_id);
/* After cafehip_set_families: every rank's block [block_lo[r], block_hi[r]) of the GLOBAL table (world entries each,
 * contiguous, starting on multiples of CAFEHIP_CHUNK); this rank's table must hold its block's rows.  Sizes the
 * exchange.  Call again whenever a table is loaded. */
int cafehip_comm_set_blocks(cafehip_ctx *ctx, const int32_t *block_lo, const int32_t *block_hi);
/* One objective evaluation of the sharded table == cafehip_eval_posterior of the whole table: *score and
 * *first_zero_global (index into the global table, -1 if none) are the same on every rank and bit-identical to the
 * single-GPU call. */
int cafehip_eval_posterior_sharded(cafehip_ctx *ctx, const double *node_lambda, const double *node_mu,
                                   const double *prior, double *score, int32_t *first_zero_global);
/* A peer that does not deliver its row is waited for in slices: the score kernel itself waits at most ~1 s, then the
 * host re-polls with a one-workgroup kernel until CAFEHIP_COMM_TIMEOUT_S (default 120 s) is over and the call fails.
 * Collective re-alignment after such a failure (or after one rank skipped an evaluation): every rank calls this
 * between evaluations; exchange buffers are cleared and the sequence numbers restart, as cafehip_comm_set_blocks does. */
int cafehip_comm_resync(cafehip_ctx *ctx);
/* What the set-up found, for logs and bench records (none of it is needed to run):
 *   out[0] world            out[1] mode agreed at init (2 direct, 1 rccl)   out[2] mode in use now (option "comm" may
 *   override)               out[3] 1 if the probe verdict was "direct" on every rank
 *   out[4] peer buffers this rank mapped (itself included)   out[5] peers whose probe store arrived here (-1: no probe)
 *   out[6] 1 if an RCCL communicator is live (ncclCommInitRank succeeded on every rank)
 *   out[7] ncclCommCount of that communicator (0 without one)   out[8] 1 if CAFEHIP_COMM_INJECT muted this rank (tests)
 *   out[9] host-paced re-polls so far (a peer was more than one wait slice late)
 * *probe_ms: wall time of the probe on this rank. */
#define CAFEHIP_COMM_STATUS_WORDS 16
int cafehip_comm_status(cafehip_ctx *ctx, int32_t out[CAFEHIP_COMM_STATUS_WORDS], double *probe_ms);
/* For a launcher whose ranks were killed: removes the shared-memory names this id maps to (control segment and
 * per-call gather segments); returns how many it removed.  Not collective. */
int cafehip_comm_cleanup(const void *unique_id);

/* All-gather of host blocks (report phase: Monte-Carlo null by root size, per-family p-values): rank r contributes
 * nbytes_mine <= nbytes_slot bytes, `all` receives world slots of nbytes_slot bytes in rank order. */
int cafehip_comm_allgather(cafehip_ctx *ctx, const void *mine, size_t nbytes_mine, void *all, size_t nbytes_slot);
/* rank / world of the context (0 / 1 without a communicator), the exchange mode in use (0 none, 1 rccl, 2 direct),
 * with timing enabled the duration of the last RCCL exchange (all-gather + pick-up, HIP events on the stream; the
 * direct exchange is part of the score kernel: cafehip_last_kernel_ms()[2]), the host time spent inside exchange
 * steps and the number of sharded evaluations.  Any pointer may be NULL. */
int cafehip_comm_info(cafehip_ctx *ctx, int *rank, int *world, int *mode, double *exchange_ms, double *host_seconds,
                      long *calls);

/* Test hook: the HOST half of a communicator alone (rendezvous by id, three barriers, one all-gather of host blocks,
 * as cafehip_comm_allgather) without a context or a device -- what the CPU test suite runs with several processes. */
int cafehip_comm_host_selftest(int rank, int world, const void *unique_id, const void *mine, size_t nbytes_mine,
                               void *all, size_t nbytes_slot);

/* Test hook: the mode agreement of cafehip_comm_init alone, over the same mailboxes, with this rank's local outcomes
 * injected (my_probe_ok: "my probe heard every peer"; my_rccl_ok: "I could join the RCCL communicator").  *mode = 2
 * direct, 1 rccl, 0 none -- the same on every rank.  No device. */
int cafehip_comm_mode_selftest(int rank, int world, const void *unique_id, int my_probe_ok, int my_rccl_ok, int *mode);

/* Multi-GPU exchange helper: bring `nbytes` (a multiple of 8, <= 1 MiB) of device memory -- the output of the
 * caller's collective, enqueued on the context's stream -- to the host without a copy command or a stream
 * synchronisation: a one-workgroup kernel on the same stream stores the words into the context's pinned host
 * mirror and then a sequence number, which the host polls.  *host_ptr stays valid until the next call.
 * (No reference counterpart: the reference's reduction is a host loop, cafe/lambda.cpp:698-722.) */
int cafehip_fetch_small(cafehip_ctx *ctx, const void *d_src, size_t nbytes, const void **host_ptr);

/* Test hook (host only): the two restated forms of the host libm's exp() (cafe_amd/csrc/exp_like_host.hpp) against this
 * host's exp() on n pseudo-random arguments; returns the form K1's exact arithmetic uses for the report phase's matrices
 * (1: the fused-multiply-add build, 2: the plain build, 0: neither matched -- the device library's exp), and how many
 * arguments each form missed.  Replaces nothing: it is how "the same bits as the reference's build on this machine" is kept
 * for the comparisons of cafe/viterbi.cpp:60-67 and cafe/cafe_tree.c:533-569. */
int cafehip_exp_like_host_selftest(long n, unsigned seed, long *mismatches_fused, long *mismatches_plain);

/* Timing of the kernels of the last cafehip_eval_posterior call, measured with HIP
 * events on the context's stream: ms[0] = matrix build, ms[1] = pruning+posterior,
 * ms[2] = score reduction.  Enabled by cafehip_enable_timing(ctx, 1). */
int cafehip_enable_timing(cafehip_ctx *ctx, int on);
int cafehip_last_kernel_ms(cafehip_ctx *ctx, double ms[3]);
/* The part of ms[1] spent building the factor tables of compressed subtrees (the k2c_nodes launches that precede the
 * family walk; 0 when the table does not compress): ms[1] - this = the walk launch alone. */
int cafehip_last_tables_ms(cafehip_ctx *ctx, double *ms);
/* With timing enabled: duration of the pruning launch of the last cafehip_eval_root_likelihoods call (HIP
 * events on the context's stream; the copies either side of it are not included). */
int cafehip_last_batch_ms(cafehip_ctx *ctx, double *ms);

/* Human-readable description of the last launch geometry (for logs/tests). */
const char *cafehip_describe(cafehip_ctx *ctx);

const char *cafehip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
