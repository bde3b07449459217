/*
 * cafe_oracle.h -- CPU restatement of CAFE's per-family likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product path
 * (cafe_amd/, libcafehip.so) never links, imports or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"): the restatement is checked against
 *   - the known-answer values of the reference's own unit tests
 *     (tests/test.cpp, tests/lambda_tests.cpp; table in SURVEY.md section 4),
 *   - the golden transcripts tests/integration/test1.t and test2.t
 *     (lambda -> score pairs over 14,787 and 4 families),
 *   - reference outputs recorded in SURVEY.md section 8(c),
 *   - a partial build of the reference's own sources that compile directly
 *     (oracle/_ref: libcommon/mathfunc.c, libtree/chooseln_cache.c,
 *     libcommon/fminsearch.cpp, ...) for gammaln / chooseln / pvalue /
 *     Nelder-Mead bit-level comparison.
 * libtree/birthdeath.c and cafe/ need the autoconf-generated config.h, which
 * this image cannot generate, so the full reference is unbuildable here.
 *
 * All citations are path:line relative to the reference tree.
 */
#ifndef CAFE_ORACLE_H
#define CAFE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* libtree/family.h:10-15 */
typedef struct {
    int min;
    int max;
    int root_min;
    int root_max;
} orc_range;

/* A tree in the reference's nlist (in-order) numbering
 * (cafe/cafe_commands.cpp:2028-2051): node ids 0..n_nodes-1, leaves have
 * left == right == -1, root has parent == -1. */
typedef struct {
    int n_nodes;
    const int *parent;
    const int *left;
    const int *right;
    const double *branchlength;
    int root;
} orc_tree;

/* ---- libcommon/mathfunc.c ------------------------------------------------ */
double orc_gammaln(double a);                                 /* :112-119 */
double orc_chooseln(double n, double r);                      /* :224-229 */
double orc_poisspdf(int x, double lambda);                    /* :352-355 */
double orc_pvalue(double v, const double *conddist, int size);/* :663-689 */
int    orc_maxidx(const double *data, int size);              /* :9-24   */
double orc_max(const double *data, int size);                 /* :26-40  */
double orc_unifrnd(void);                                     /* :91-94  */

/* ---- libtree/chooseln_cache.{h,c} ---------------------------------------- */
/* Dense table T[n*(size+1)+x] = chooseln(n,x) for n < 2*size, x <= min(n,size);
 * entries with x > n are NaN (never read by the BD sum).  Caller frees. */
double *orc_chooseln_table(int size);

/* ---- libtree/birthdeath.c ------------------------------------------------ */
double orc_birthdeath_rate_with_log_alpha(int s, int c, double log_alpha, double coeff,
                                          const double *lnc, int size);            /* :52-73 */
double orc_birthdeath_rate_with_log_alpha_beta(int s, int c, double log_alpha, double log_beta,
                                               double log_coeff, const double *lnc, int size); /* :34-50 */
double orc_birthdeath_likelihood_with_s_c(int s, int c, double branchlength, double lambda,
                                          double mu, const double *lnc, int size); /* :80-119 */
/* out: (M+1)*(M+1) row-major, out[s*(M+1)+c] = Pr(c | s).  :238-286 */
void orc_compute_birthdeath_rates(double branchlength, double lambda, double mu, int M,
                                  double *out);
void orc_square_matrix_multiply(const double *m, int size, const double *v, int row_start,
                                int row_end, int col_start, int col_end, double *result); /* :163-182 */

/* ---- cafe/cafe_tree.c: matrix cache + pruning ---------------------------- */
/* Per-node matrices for given per-node (lambda, mu); keys are (int)branchlength,
 * lambda, mu (libtree/birthdeath.h:26-31, cafe/cafe_tree.c:374-391, 461-483).
 * Returns a handle; mats[node] points at the node's S*S matrix (NULL for the root
 * or branchlength <= 0). */
typedef struct orc_matrices orc_matrices;
orc_matrices *orc_matrices_build(const orc_tree *t, const double *node_lambda,
                                 const double *node_mu, int M, int nthreads);
int orc_matrices_nkeys(const orc_matrices *m);
int orc_matrices_size(const orc_matrices *m);
const double *orc_matrices_get(const orc_matrices *m, int node);
void orc_matrices_free(orc_matrices *m);

/* compute_tree_likelihoods (cafe/cafe_tree.c:191-323) for ONE family.
 * familysize[n_nodes]: leaf counts at leaf nodes (ignored elsewhere).
 * errormatrix: NULL or (err_mfs+1)^2 row-major [observed][true]; leaf_has_err[n_nodes].
 * L: n_nodes * sof doubles of scratch, sof = size_of_factor >= max(R, C).
 * On return L[root*sof + i], i < R, is the root likelihood for root size root_min+i. */
void orc_compute_tree_likelihoods(const orc_tree *t, const orc_range *range,
                                  const orc_matrices *mats, const int *familysize,
                                  const double *errormatrix, int err_mfs,
                                  const unsigned char *leaf_has_err, double *L, int sof);

/* compute_posterior (cafe/lambda.cpp:657-689) on a root likelihood vector. */
void orc_compute_posterior(const double *likelihood, int rfsize, const double *prior,
                           double *max_likelihood, int *argmax, double *max_posterior);

/* reset_birthdeath_cache + get_posterior (cafe/cafe_main.c:319-326,
 * cafe/lambda.cpp:691-724) == one objective evaluation.
 * counts: F x n_leaves int32, leaf slot j <-> node 2*j.  ref: NULL or F entries
 * (cafe/cafe_family.c:9-34).  Outputs may be NULL.  Returns the score
 * (-inf when some family has max_lik == 0; *first_zero_family = its index, else -1).
 * nthreads > 1 parallelises over families with OpenMP (baseline timing only; the
 * per-family values are identical, the score is still summed in family order). */
double orc_eval_posterior(const orc_tree *t, int F, int n_leaves, const int *counts,
                          const int *ref, const orc_range *range,
                          const double *node_lambda, const double *node_mu,
                          const double *prior, const double *errormatrix, int err_mfs,
                          const unsigned char *leaf_has_err, int nthreads,
                          int *first_zero_family, double *max_lik, int *argmax_root,
                          double *max_post);

/* Root likelihood vectors for a batch of families with per-row extents, as the
 * MC null and the report phase use them (cafe/conditional_distribution.cpp:10-44,
 * cafe/cafe_family.c:236-255).  out is packed: sum(root_hi-root_lo+1) doubles. */
void orc_eval_root_likelihoods(const orc_tree *t, int B, int n_leaves, const int *counts,
                               const int *root_lo, const int *root_hi, const int *col_max,
                               const orc_matrices *mats, double *out);

/* ---- k-cluster model: cafe/cafe_main.c:165-253, cafe/cafe_tree.c:704-850 ---- */
void orc_copy_weights(double *out, const double *parameters, int start, int count);  /* libtree/input_values.c:84-94 */
double orc_eval_clustered_posterior(const orc_tree *t, int F, int n_leaves, const int *counts, const int *ref,
                                    const orc_range *range, int K, const double *node_lambda,
                                    const double *node_mu, const double *weights, const double *prior,
                                    int nthreads, double *MAP_out, double *p_z_out, double *new_weights,
                                    int *first_zero_family);

/* ---- cafe/cafe_family.c -------------------------------------------------- */
void orc_init_family_size(orc_range *fs, int max);            /* :357-364 */
void orc_family_check_the_pattern(int F, int n, const int *counts, int *ref); /* :9-34 */

/* ---- cafe/lambda.cpp: prior ---------------------------------------------- */
/* prior[i] = poisspdf(shift-1+i, lambda), i < n  (:841-852) */
void orc_prior_poisson(double *prior, int n, int shift, double lambda);
/* -sum log poisspdf(x, lambda) over x = count-1 for positive counts (:771-806) */
double orc_lnLPoisson(double lambda, int F, int n, const int *counts);
/* find_poisson_lambda (:808-838) from start x0; returns lambda, *iters, *score */
double orc_find_poisson_lambda(int F, int n, const int *counts, double x0, int *iters,
                               double *score);

/* ---- libcommon/fminsearch.cpp -------------------------------------------- */
typedef double (*orc_math_func)(double *x, void *args);
/* Nelder-Mead exactly as fminsearch_min (:264-302) with the defaults of
 * fminsearch_new (:7-21).  x0[N] in, xmin[N] out; returns iterations. */
int orc_fminsearch(orc_math_func eq, int N, void *args, const double *x0, double tolx,
                   double tolf, int maxiters, double *xmin, double *fmin, int *bymax);

/* ---- cafe/lambda.cpp:726-769 objective + search -------------------------- */
/* Objective for x[num_params] with node_class[n_nodes] (taxaid from the lambda
 * tree, 0 if none); has_mu selects the lambdamu layout (cafe/cafe_shell.c:31-38,
 * 148-177, 46-146 without eqbg/k): lambda = x[class], mu = x[num_lambdas+class]. */
typedef struct {
    const orc_tree *t;
    int F, n_leaves;
    const int *counts;
    const int *ref;
    orc_range range;
    const double *prior;
    const int *node_class;
    int num_lambdas;
    int has_mu;
    int nthreads;
    int n_evals;           /* out: number of objective calls */
    /* optional trace of (x[0..], score) per call */
    double *trace;         /* NULL or capacity trace_cap*(num_params+1) */
    int trace_cap;
} orc_objective;
double orc_lambda_objective(double *x, void *args);            /* returns -score */

/* ---- Monte-Carlo null (cafe/conditional_distribution.cpp, cafe_tree.c:533-569) */
/* Uses libc rand() exactly as the reference (-t 1 order).  out: R x trials, each row
 * sorted ascending. mats must be built with M = max(range.max, range.root_max). */
void orc_conditional_distribution(const orc_tree *t, const orc_range *range,
                                  const orc_matrices *mats, int trials, double *out);
/* One simulated family: fills familysize[n_nodes]; returns max size. */
int orc_tree_random_familysize(const orc_tree *t, const orc_matrices *mats, int root_size,
                               int max_family_size, int *familysize);

/* ---- Viterbi / p-values (cafe/viterbi.cpp, cafe/pvalue.cpp) ------------------------ */
/* cafe_tree_viterbi (cafe/viterbi.cpp:494-516): max-product pass (:208-320) + backtrack (:322-351)
 * for ONE family under `range`.  familysize[n_nodes]: leaf counts in, all node sizes out.
 * vit: n_nodes * sof ints of persistent state (the reference never clears node->viterbi, so rows
 * whose products are all zero keep whatever an earlier family left there); L: n_nodes*sof scratch. */
void orc_tree_viterbi(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                      int *familysize, int *vit, double *L, int sof);
/* cafe_tree_p_values (cafe/pvalue.cpp:143-154): pvalues[s] for s < rfsize against cd (R x trials). */
void orc_tree_p_values(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                       const int *familysize, const double *cd, int trials, double *pvalues);
/* viterbi_sum_probabilities (cafe/viterbi.cpp:44-71): out[2*j+k] for internal node 2j+1, child k. */
void orc_viterbi_sum_probabilities(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                                   const int *familysize, double *out);
/* cafe_family_set_size_with_family_forced ranges (cafe/cafe_family.c:236-255) */
void orc_family_forced_range(orc_range *r, int n_leaves, const int *row);

#ifdef __cplusplus
}
#endif
#endif
