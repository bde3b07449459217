/*
 * cafehost.h -- C API of the host-side driver that sits ABOVE the kernel boundary
 * (include/cafehip.h): CAFE's command language for the hot path -- `seed`, `load`,
 * `tree`, `lambda`, `lambdamu`, `errormodel` -- with the reference's Nelder-Mead
 * search, parameter scatter, prior fit and log lines, every objective evaluation
 * running on the GPU through cafehip_eval_posterior.
 *
 * Reference interfaces mirrored (names kept where a function exists):
 *   command dispatcher            cafe/cafe_commands.cpp:174-217, 504-536
 *   load / tree / seed            cafe/cafe_commands.cpp:817-866, 1127-1190, 1950-1965
 *   lambda / lambdamu             cafe/lambda.cpp:369-515, cafe/lambdamu.cpp:218-269
 *   cafe_best_lambda_by_fminsearch cafe/lambda.cpp:525-647; fminsearch_min libcommon/fminsearch.cpp:264-302
 *   __cafe_best_lambda_search     cafe/lambda.cpp:726-769; cafe_best_lambda_mu_search cafe/lambdamu.cpp:323-367
 *   cafe_shell_set_lambdas        cafe/cafe_shell.c:31-38, 148-177, 46-146
 *   cafe_set_prior_rfsize_empirical cafe/lambda.cpp:808-870
 */
#ifndef CAFEHOST_H
#define CAFEHOST_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cafehost_session cafehost_session;

/* One session == the reference's global CafeParam (cafe/cafe_shell.c:20) bound to HIP device
 * `device_id`.  Log lines go to `log_path` ("stdout" or NULL = stdout). */
int cafehost_create(cafehost_session **out, int device_id, const char *log_path);
void cafehost_destroy(cafehost_session *s);

/* Execute one command line exactly as the reference's REPL would read it
 * (cafe_shell_dispatch_command, cafe/cafe_commands.cpp:504-536).  Returns 0, or <0 with the
 * message in cafehost_last_error(); the `exit` command returns 1. */
int cafehost_dispatch(cafehost_session *s, const char *command_line);

/* Run a script file (main.cpp:43 `source`): one command per line, '#' lines ignored. */
int cafehost_run_script(cafehost_session *s, const char *path);

/* Test hook (no session, no GPU): the Monte-Carlo null draws its ~15 million uniforms in one loop over the state of
 * glibc's default generator instead of 15 million random_r calls; returns 0 when n_before single draws, n_bulk bulk
 * draws and n_after single draws after srandom(seed) reproduce random_r's own stream (the reference's rand(),
 * libcommon/mathfunc.c unifrnd). */
int cafehost_rng_selftest(unsigned seed, int n_before, int n_bulk, int n_after);
/* Test hooks (no session, no GPU) for the two host routines whose exact behaviour steers results: the rank of a
 * likelihood in a sorted Monte-Carlo null (pvalue, libcommon/mathfunc.c:663-689) and the Nelder-Mead of the searches
 * (fminsearch_min, libcommon/fminsearch.cpp:264-302: returns the iteration count, fills the best vertex / value).
 * tests/test_host_vs_ref_build.py compares both bit for bit with the reference's own objects (oracle/_ref). */
/* The empirical root-size prior's Poisson fit (find_poisson_lambda, cafe/lambda.cpp:771-838) on the given leaf sizes
 * (count - 1 of every non-zero count, table order) from the given start; lookahead = 0: one sweep over the sizes per
 * objective call as the reference does it, 1: the points Nelder-Mead may ask for evaluated several per sweep -- the
 * fitted value, the score and the iteration count must be the same bits.  *passes = sweeps over the table. */
int cafehost_poisson_fit_selftest(const int *leaf_sizes, long n, double start, int lookahead, double *lambda, double *score,
                                  int *iters, long *passes);
typedef double (*cafehost_math_fn)(double *x, void *args);
double cafehost_pvalue_selftest(double v, const double *sorted_null, int size);
/* Test hook: the number formatting of the report's per-family lines (std::to_chars, general, precision 6 / integers)
 * against printf("%g") / printf("%d") on the caller's values; returns the number that differ (0 expected). */
long cafehost_format_selftest(const double *values, long n, double *first_bad);
int cafehost_fminsearch_selftest(cafehost_math_fn eq, int n, void *args, const double *x0, double tolx, double tolf,
                                 double *xmin, double *fmin, int *bymax);
/* Test hook: the same minimisation with the look-ahead hook installed (FMinSearch::lookahead, the points announced before
 * each evaluation for the one after it).  out = {evaluations, evaluations whose point had been announced -- compared bit for
 * bit --, announcements, points announced}; the trajectory must be the plain one's. */
int cafehost_lookahead_selftest(cafehost_math_fn eq, int n, void *args, const double *x0, double tolx, double tolf,
                                double *xmin, double *fmin, long out[4]);

/* ---- multi-GPU (one process per GPU) ------------------------------------------------------------
 * Every rank runs the same script (same seed => same Nelder-Mead decisions); a rank scores only its
 * chunk-aligned block of the family table and the ranks exchange the per-chunk partial sums once per
 * objective call.  Two ways to provide that exchange: the native communicator below (cafehost_init_comm: RCCL
 * behind this ABI, nothing for the caller to do), or -- for a caller that already owns a process group, e.g.
 * torch.distributed in cafe_amd/multi_gpu.py -- a callback: the driver fills the caller's device buffers
 * asynchronously and then calls `exchange`, which must return the global score and set *first_zero_global
 * (< 0 if none). */
typedef double (*cafehost_exchange_fn)(void *user, int *first_zero_global);
int cafehost_set_shard(cafehost_session *s, int rank, int world);
int cafehost_shard_bounds(cafehost_session *s, int *lo, int *hi, int *n_chunks_local);
int cafehost_set_exchange(cafehost_session *s, cafehost_exchange_fn exchange, void *user,
                          void *d_chunk_sums, void *d_first_zero);
/* `report` and `pvalue` (Monte-Carlo null, per-family root likelihoods, Viterbi): with an allgather callback
 * registered the simulated families are sharded by root size (as the reference's threads are,
 * cafe/conditional_distribution.cpp:88-108) and the observed families by contiguous block; the callback
 * receives this rank's `nbytes_mine` bytes and must fill `all` with world slots of `nbytes_slot` bytes each
 * (rank order, own data first in its slot).  Random draws stay on the host in the reference's global
 * order, so the output does not depend on the number of ranks.  Only rank 0 writes files.  Without a
 * callback every rank computes everything. */
typedef int (*cafehost_allgather_fn)(void *user, const void *mine, long long nbytes_mine, void *all,
                                     long long nbytes_slot);
int cafehost_set_allgather(cafehost_session *s, cafehost_allgather_fn fn, void *user);
/* ---- native communicator: the exchange behind the C ABI, no callbacks ----------------------------------------
 * Rank 0 obtains an id and hands its CAFEHOST_COMM_ID_BYTES bytes to the other ranks by any means (file, pipe,
 * environment; cafe_amd/bin/cafehip --gpus N writes it to a temporary file); every rank then calls
 * cafehost_init_comm with its rank.  From then on the session shards every table it loads (contiguous,
 * chunk-aligned blocks) and every objective evaluation is cafehip_eval_posterior_sharded (include/cafehip.h): the
 * ranks exchange one packed row each -- directly between the score kernels over xGMI, or with one ncclAllGather
 * (option "comm") -- the map + sum of cafe/lambda.cpp:698-722, followed by the same fixed-order sum on every rank,
 * so every rank's Nelder-Mead takes identical decisions and the score does not depend on the number of GPUs.
 * `report`, `pvalue` and `lhtest` run sharded too (the Monte-Carlo null by root size, as
 * cafe/conditional_distribution.cpp:88-108 splits it over threads); rank 0 writes the files. */
#define CAFEHOST_COMM_ID_BYTES 128
int cafehost_comm_unique_id(void *out_id /* CAFEHOST_COMM_ID_BYTES */);
int cafehost_init_comm(cafehost_session *s, int rank, int world, const void *unique_id);
/* launcher only, after it had to kill its ranks: removes the shared-memory names of that id (cafehip_comm_cleanup) */
int cafehost_comm_cleanup(const void *unique_id);
/* host time inside RCCL exchange steps (collective launch + result pick-up; the direct exchange has none: it is
 * part of the score kernel) and the number of sharded evaluations */
int cafehost_exchange_stats(cafehost_session *s, double *seconds, long *calls);

int cafehost_set_stream(cafehost_session *s, void *hip_stream);
/* cafehip_fetch_small on the session's device context (for the exchange callback: collective output -> host). */
int cafehost_fetch_small(cafehost_session *s, const void *d_src, unsigned long nbytes, const void **host_ptr);
/* Upload tree + (sharded) table now instead of at the first lambda command (so that the caller can size
 * its exchange buffers from cafehost_shard_bounds). */
int cafehost_upload(cafehost_session *s);

/* Results of the last lambda / lambdamu command. */
int cafehost_num_params(cafehost_session *s);
int cafehost_get_params(cafehost_session *s, double *out, int n);   /* fitted lambda(s) [, mu(s)] */
double cafehost_last_score(cafehost_session *s);                    /* -lnL as printed ("Score") */
int cafehost_search_iterations(cafehost_session *s);
int cafehost_num_evaluations(cafehost_session *s);                  /* objective calls of the last command */
double cafehost_search_seconds(cafehost_session *s);                /* wall-clock of the last search */
double cafehost_poisson_lambda(cafehost_session *s);

/* Run-time switches: "speculate" (auto|0|1: batched candidate evaluation, below), "timing" (0|1: phase times of
 * report / the Monte-Carlo null on stderr), "prior_file" (path, "" = off: the searches take the root-size prior from
 * this file -- one probability per line for root sizes root_min, root_min + 1, ... -- instead of fitting the
 * reference's empirical Poisson, cafe/lambda.cpp:808-870; an extension for tables whose root distribution is known),
 * "report_arith" (fast|reference: the likelihood vectors of report / pvalue on the matrix cores, or in the reference's
 * own arithmetic -- exact-form matrices with the host libm's exp() and a separate multiply and add per term -- which
 * returns the oracle's bits, Monte-Carlo null included, at about 2.7x the report time), "objective_arith" (fast|reference:
 * the same for every objective evaluation of a search, the posterior and the sum of logs formed on the host in the reference's
 * order with the host's libm -- the score is the oracle's double; one GPU, no error model, slow: a proof, not a mode),
 * "prior_lookahead" (0|1: the Poisson
 * fit evaluates Nelder-Mead's candidate points several per sweep, same bits), "lookahead" (auto|0|1: matrices of the points a
 * search may ask for next built ahead of time, cafehost_lookahead_stats), "lhtest_deal" (0|1: a sharded job deals
 * lhtest's files to the ranks);
 * every other key is handed to cafehip_set_option on the session's device
 * context(s) (include/cafehip.h).  CAFEHOST_SPECULATE / CAFEHOST_TIMING in the environment are read once, by
 * cafehost_create. */
int cafehost_set_option(cafehost_session *s, const char *key, const char *value);

/* Batched candidate evaluation of the searches: passes launched, points evaluated in them, and how
 * many objective calls took their value from such a pass.  A table that fills less than half of the chip has the
 * four candidates of every Nelder-Mead iteration (and the points of a `lambda -r` grid) evaluated together through
 * cafehip_eval_posterior_multi; trajectories and log lines are those of the sequential run.
 * Option speculate=0 / 1 forces it off / on. */
int cafehost_speculation_stats(cafehost_session *s, long *launches, long *points, long *hits);

/* Matrices ahead of time (round 5): where whole evaluations are not batched, the searches announce the points Nelder-Mead
 * may ask for AFTER the evaluation it is about to start -- the expansion / contraction point of the pending reflection,
 * the reflection of every simplex the pending decision can leave behind, the vertices of a shrink (all functions of the
 * simplex, libcommon/fminsearch.cpp:189-250) -- and the library builds their transition matrices beside that evaluation's
 * pruning (cafehip_prefetch_matrices).  Call order, values, log lines: those of the plain loop.  out = {announcements,
 * points announced, evaluations that found their matrices on the device, sets built ahead of time}.
 * Option lookahead=0 / 1 forces it off / on (CAFEHOST_LOOKAHEAD in the environment, read once by cafehost_create). */
int cafehost_lookahead_stats(cafehost_session *s, long out[4]);

/* Trace of the last command's objective calls: row i = (params[0..num_params), score).
 * Returns the number of rows copied (<= max_rows). */
int cafehost_get_trace(cafehost_session *s, double *out, int max_rows);

const char *cafehost_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
