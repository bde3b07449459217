/*
 * cafe_oracle.c -- CPU restatement of CAFE's per-family likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cafe_oracle.h).  Written from the algorithm the
 * reference implements; every function cites the reference file:line it follows.
 * Compile WITHOUT fp contraction (-ffp-contract=off) so that the arithmetic is
 * the same sequence of IEEE double operations gcc emits for the reference on
 * x86-64 without -march flags.
 */
#include "cafe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ======================================================================== */
/* libcommon/mathfunc.c                                                      */
/* ======================================================================== */

/* libcommon/mathfunc.c:9-24 -- first maximum wins (strict <) */
int orc_maxidx(const double *data, int size)
{
    double max = data[0];
    int v = 0;
    for (int i = 1; i < size; i++) {
        if (max < data[i]) {
            v = i;
            max = data[i];
        }
    }
    return v;
}

/* libcommon/mathfunc.c:26-40 */
double orc_max(const double *data, int size)
{
    double max = data[0];
    for (int i = 1; i < size; i++) {
        if (max < data[i]) max = data[i];
    }
    return max;
}

/* libcommon/mathfunc.c:91-94 */
double orc_unifrnd(void)
{
    return rand() / (RAND_MAX + 1.0);
}

/* Lanczos coefficients, libcommon/mathfunc.c:87-89 */
static const double Qs[7] = {1.000000000190015,  76.18009172947146,    -86.50532032941677,
                             24.01409824083091,  -1.231739572450155,   1.208650973866179e-3,
                             -5.395239384953e-6};
#define ORC_SQRT_2PI 2.5066282746310002416123552393401042 /* libcommon/mathfunc.c:105 */

/* libcommon/mathfunc.c:112-119 */
double orc_gammaln(double a)
{
    double p = Qs[0];
    double a_add_5p5 = a + 5.5;
    for (int n = 1; n <= 6; n++) p += Qs[n] / (a + n);
    return (a + 0.5) * log(a_add_5p5) - (a_add_5p5) + log(ORC_SQRT_2PI * p / a);
}

/* libcommon/mathfunc.c:224-229 */
double orc_chooseln(double n, double r)
{
    if (r == 0 || (n == 0 && r == 0)) return 0;
    else if (n <= 0 || r <= 0) return log(0);
    return orc_gammaln(n + 1) - orc_gammaln(r + 1) - orc_gammaln(n - r + 1);
}

/* libcommon/mathfunc.c:352-355 */
double orc_poisspdf(int x, double lambda)
{
    return exp(x * log(lambda) - orc_gammaln(x + 1) - lambda);
}

/* libcommon/mathfunc.c:663-689 */
double orc_pvalue(double v, const double *conddist, int size)
{
    int from = 0;
    int to = size - 1;
    int mi;
    while (from < to) {
        mi = from + (to - from) / 2;
        if (conddist[mi] > v) {
            to = mi - 1;
        } else if (conddist[mi] < v) {
            from = mi + 1;
        } else {
            for (from = mi - 1; from >= 0 && conddist[from] == v; from--)
                ;
            for (to = mi + 1; to < size && conddist[to] == v; to++)
                ;
            from++, to--;
            break;
        }
    }
    if (from > to) to = from;
    return (double)(from + (conddist[from] <= v ? 1 : 0) + (to - from) / 2.0) / (double)size;
}

/* ======================================================================== */
/* libtree/chooseln_cache.{h,c}                                              */
/* ======================================================================== */

/* The reference fills cache->values[n][x] = chooseln(n, x) lazily
 * (libtree/chooseln_cache.h:27-41) for n < 2*size, x <= size and pre-touches
 * exactly the pairs the BD sum reads (libtree/chooseln_cache.c:16-34).  The
 * cached value is a pure function of (n, x), so a dense table is value-identical. */
double *orc_chooseln_table(int size)
{
    int rows = 2 * size;
    int ld = size + 1;
    if (rows < 2) rows = 2;
    double *T = (double *)malloc(sizeof(double) * (size_t)rows * ld);
    for (int n = 0; n < rows; n++) {
        for (int x = 0; x <= size; x++) {
            T[(size_t)n * ld + x] = (x <= n) ? orc_chooseln(n, x) : NAN;
        }
    }
    return T;
}

/* ======================================================================== */
/* libtree/birthdeath.c                                                      */
/* ======================================================================== */

/* libtree/birthdeath.c:52-73 */
double orc_birthdeath_rate_with_log_alpha(int s, int c, double log_alpha, double coeff,
                                          const double *lnc, int size)
{
    int ld = size + 1;
    int m = ORC_MIN(c, s);
    double lastterm = 1;
    double p = 0.0;
    int s_add_c = s + c;
    int s_add_c_sub_1 = s_add_c - 1;
    int s_sub_1 = s - 1;
    for (int j = 0; j <= m; j++) {
        double t = lnc[(size_t)s * ld + j] + lnc[(size_t)(s_add_c_sub_1 - j) * ld + s_sub_1] +
                   (s_add_c - 2 * j) * log_alpha;
        p += (exp(t) * lastterm);
        lastterm *= coeff;
    }
    return ORC_MAX(ORC_MIN(p, 1), 0);
}

/* libtree/birthdeath.c:34-50 */
double orc_birthdeath_rate_with_log_alpha_beta(int s, int c, double log_alpha, double log_beta,
                                               double log_coeff, const double *lnc, int size)
{
    int ld = size + 1;
    int m = ORC_MIN(c, s);
    double t, p = 0;
    int s_add_c = s + c;
    int s_add_c_sub_1 = s_add_c - 1;
    int s_sub_1 = s - 1;
    for (int j = 0; j <= m; j++) {
        t = lnc[(size_t)s * ld + j] + lnc[(size_t)(s_add_c_sub_1 - j) * ld + s_sub_1] +
            (s - j) * log_alpha + (c - j) * log_beta + j * log_coeff;
        p += exp(t);
    }
    return ORC_MAX(ORC_MIN(p, 1), 0);
}

/* libtree/birthdeath.c:80-119 */
double orc_birthdeath_likelihood_with_s_c(int s, int c, double branchlength, double lambda,
                                          double mu, const double *lnc, int size)
{
    double alpha, coeff, beta = 0;
    double denominator, numerator = 0;
    if (s == 0) {
        return c == 0 ? 1 : 0;
    }
    if ((mu < 0) || (lambda == mu)) {
        alpha = lambda * branchlength / (1 + lambda * branchlength);
        coeff = 1 - 2 * alpha;
        if (coeff <= 0) return 0;
        return orc_birthdeath_rate_with_log_alpha(s, c, log(alpha), coeff, lnc, size);
    }
    denominator = lambda * (exp((lambda - mu) * branchlength)) - mu;
    numerator = exp((lambda - mu) * branchlength) - 1;
    alpha = (mu * numerator) / denominator;
    beta = (lambda * numerator) / denominator;
    coeff = 1 - alpha - beta;
    if (coeff <= 0) return 0;
    return orc_birthdeath_rate_with_log_alpha_beta(s, c, log(alpha), log(beta), log(coeff), lnc,
                                                   size);
}

static void compute_rates_with_table(double branchlength, double lambda, double mu, int M,
                                     const double *lnc, double *out)
{
    /* libtree/birthdeath.c:238-286; init_matrix :210-225 */
    int sz = M + 1;
    memset(out, 0, sizeof(double) * (size_t)sz * sz); /* calloc'd by square_matrix_init :121-125 */
    out[0] = 1; /* :244 once you are zero you are almost surely zero */

    double alpha = 0, beta = 0, coeff = 1;
    if (mu < 0 || lambda == mu) {
        alpha = lambda * branchlength / (1 + lambda * branchlength);
        beta = alpha;
        coeff = 1 - 2 * alpha;
    } else {
        double e_diff = exp((lambda - mu) * branchlength);
        double numerator = e_diff - 1;
        double denominator = lambda * (e_diff)-mu;
        alpha = (mu * numerator) / denominator;
        beta = (lambda * numerator) / denominator;
        coeff = 1 - alpha - beta;
    }
    /* init_matrix: row 0 cols >= 1 are 0 (already); coeff <= 0 -> rows 1.. zero;
     * coeff == 1 -> identity in rows 1.. */
    if (coeff <= 0) {
        return;
    } else if (coeff == 1) {
        for (int s = 1; s < sz; s++) out[(size_t)s * sz + s] = 1;
        return;
    }
    /* coeff > 0 && coeff != 1 */
    double log_alpha = log(alpha);
    double log_beta = log(beta);
    double log_coeff = log(coeff);
    for (int s = 1; s <= M; s++) {
        for (int c = 0; c <= M; c++) {
            if (mu < 0)
                out[(size_t)s * sz + c] =
                    orc_birthdeath_rate_with_log_alpha(s, c, log_alpha, coeff, lnc, M);
            else
                out[(size_t)s * sz + c] = orc_birthdeath_rate_with_log_alpha_beta(
                    s, c, log_alpha, log_beta, log_coeff, lnc, M);
        }
    }
}

void orc_compute_birthdeath_rates(double branchlength, double lambda, double mu, int M,
                                  double *out)
{
    double *lnc = orc_chooseln_table(M);
    compute_rates_with_table(branchlength, lambda, mu, M, lnc, out);
    free(lnc);
}

/* libtree/birthdeath.c:163-182 (non-BLAS branch) */
void orc_square_matrix_multiply(const double *m, int size, const double *v, int row_start,
                                int row_end, int col_start, int col_end, double *result)
{
    for (int s = row_start, i = 0; s <= row_end; s++, i++) {
        result[i] = 0;
        for (int c = col_start, j = 0; c <= col_end; c++, j++) {
            result[i] += m[(size_t)s * size + c] * v[j];
        }
    }
}

/* ======================================================================== */
/* matrix cache: cafe/cafe_tree.c:339-483, libtree/birthdeath.h:26-31        */
/* ======================================================================== */

struct orc_matrices {
    int n_nodes;
    int size; /* S = M + 1 */
    int nkeys;
    int *key_bl;
    double *key_lambda;
    double *key_mu;
    double *storage;   /* nkeys * S * S */
    int *node_key;     /* -1 if none */
};

orc_matrices *orc_matrices_build(const orc_tree *t, const double *node_lambda,
                                 const double *node_mu, int M, int nthreads)
{
    orc_matrices *m = (orc_matrices *)calloc(1, sizeof(*m));
    int n = t->n_nodes;
    m->n_nodes = n;
    m->size = M + 1;
    m->key_bl = (int *)malloc(sizeof(int) * n);
    m->key_lambda = (double *)malloc(sizeof(double) * n);
    m->key_mu = (double *)malloc(sizeof(double) * n);
    m->node_key = (int *)malloc(sizeof(int) * n);
    m->nkeys = 0;
    for (int i = 0; i < n; i++) {
        m->node_key[i] = -1;
        /* cafe/cafe_tree.c:341-342, 395-396: nodes with branchlength <= 0 get no matrix
         * (the root's branchlength is -1, libtree/phylogeny.c) */
        if (i == t->root || !(t->branchlength[i] > 0)) continue;
        int bl = (int)t->branchlength[i]; /* cafe/cafe_tree.c:376 */
        int k;
        for (k = 0; k < m->nkeys; k++) {
            if (m->key_bl[k] == bl && m->key_lambda[k] == node_lambda[i] &&
                m->key_mu[k] == node_mu[i])
                break; /* cafe/cafe_tree.c:380-382 exact-double equality */
        }
        if (k == m->nkeys) {
            m->key_bl[k] = bl;
            m->key_lambda[k] = node_lambda[i];
            m->key_mu[k] = node_mu[i];
            m->nkeys++;
        }
        m->node_key[i] = k;
    }
    size_t ss = (size_t)m->size * m->size;
    m->storage = (double *)malloc(sizeof(double) * ss * (size_t)ORC_MAX(m->nkeys, 1));
    double *lnc = orc_chooseln_table(M);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int k = 0; k < m->nkeys; k++) {
        /* cafe/cafe_tree.c:468-476: compute_birthdeath_rates(key->branchlength (int!), ...) */
        compute_rates_with_table((double)m->key_bl[k], m->key_lambda[k], m->key_mu[k], M, lnc,
                                 m->storage + ss * k);
    }
    free(lnc);
    return m;
}

int orc_matrices_nkeys(const orc_matrices *m) { return m->nkeys; }
int orc_matrices_size(const orc_matrices *m) { return m->size; }

const double *orc_matrices_get(const orc_matrices *m, int node)
{
    if (node < 0 || node >= m->n_nodes || m->node_key[node] < 0) return NULL;
    return m->storage + (size_t)m->size * m->size * m->node_key[node];
}

void orc_matrices_free(orc_matrices *m)
{
    if (!m) return;
    free(m->key_bl);
    free(m->key_lambda);
    free(m->key_mu);
    free(m->node_key);
    free(m->storage);
    free(m);
}

/* ======================================================================== */
/* pruning: cafe/cafe_tree.c:191-323                                         */
/* ======================================================================== */

typedef struct {
    const orc_tree *t;
    const orc_range *range;
    const orc_matrices *mats;
    const int *familysize;
    const double *errormatrix;
    int err_mfs;
    const unsigned char *leaf_has_err;
    double *L;
    int sof;
    double *left_factor;
    double *right_factor;
} prune_ctx;

static void prune_node(prune_ctx *cx, int node)
{
    const orc_tree *t = cx->t;
    double *Lv = cx->L + (size_t)node * cx->sof;
    if (t->left[node] < 0) {
        /* initialize_leaf_likelihoods cafe/cafe_tree.c:191-211 */
        memset(Lv, 0, sizeof(double) * cx->sof);
        if (cx->errormatrix && cx->leaf_has_err && cx->leaf_has_err[node]) {
            int ld = cx->err_mfs + 1;
            int fs = cx->familysize[node];
            for (int j = 0; j < cx->sof; j++) {
                /* reference reads errormatrix[familysize][j] for j < size_of_factor; entries
                 * beyond the matrix are never used by the mat-vec (cols <= range.max <= mfs) */
                Lv[j] = (j < ld) ? cx->errormatrix[(size_t)fs * ld + j] : 0.0;
            }
        } else {
            int fs = cx->familysize[node];
            if (fs >= 0 && fs < cx->sof) Lv[fs] = 1;
        }
        return;
    }
    /* compute_node_likelihoods_recursive :289-318: left subtree, right subtree, then node */
    prune_node(cx, t->left[node]);
    prune_node(cx, t->right[node]);

    /* compute_internal_node_likelihood :226-271 */
    int row_lo, row_hi;
    if (node == t->root) {
        row_lo = cx->range->root_min;
        row_hi = cx->range->root_max;
    } else {
        row_lo = cx->range->min;
        row_hi = cx->range->max;
    }
    int S = orc_matrices_size(cx->mats);
    const double *ma = orc_matrices_get(cx->mats, t->left[node]);
    const double *mb = orc_matrices_get(cx->mats, t->right[node]);
    /* compute_child_factor :213-224 */
    orc_square_matrix_multiply(ma, S, cx->L + (size_t)t->left[node] * cx->sof, row_lo, row_hi,
                               cx->range->min, cx->range->max, cx->left_factor);
    orc_square_matrix_multiply(mb, S, cx->L + (size_t)t->right[node] * cx->sof, row_lo, row_hi,
                               cx->range->min, cx->range->max, cx->right_factor);
    int size = row_hi - row_lo + 1;
    for (int i = 0; i < size; i++) Lv[i] = cx->left_factor[i] * cx->right_factor[i];
}

/* the walk with the caller's two factor buffers (size_of_factor doubles each): the family loops below keep one pair per
 * thread, so that a family costs no allocator call (the reference allocates its factors once per tree, cafe/cafe_tree.c:46-67) */
static void tree_likelihoods_ws(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                                const int *familysize, const double *errormatrix, int err_mfs,
                                const unsigned char *leaf_has_err, double *L, int sof, double *left_factor,
                                double *right_factor)
{
    prune_ctx cx;
    cx.t = t;
    cx.range = range;
    cx.mats = mats;
    cx.familysize = familysize;
    cx.errormatrix = errormatrix;
    cx.err_mfs = err_mfs;
    cx.leaf_has_err = leaf_has_err;
    cx.L = L;
    cx.sof = sof;
    cx.left_factor = left_factor;
    cx.right_factor = right_factor;
    memset(left_factor, 0, sizeof(double) * (size_t)sof);
    memset(right_factor, 0, sizeof(double) * (size_t)sof);
    prune_node(&cx, t->root);
}

void orc_compute_tree_likelihoods(const orc_tree *t, const orc_range *range,
                                  const orc_matrices *mats, const int *familysize,
                                  const double *errormatrix, int err_mfs,
                                  const unsigned char *leaf_has_err, double *L, int sof)
{
    double *lf = (double *)malloc(sizeof(double) * (size_t)sof);
    double *rf = (double *)malloc(sizeof(double) * (size_t)sof);
    tree_likelihoods_ws(t, range, mats, familysize, errormatrix, err_mfs, leaf_has_err, L, sof, lf, rf);
    free(lf);
    free(rf);
}

/* cafe/lambda.cpp:657-689 */
void orc_compute_posterior(const double *likelihood, int rfsize, const double *prior,
                           double *max_likelihood, int *argmax, double *max_posterior)
{
    *max_likelihood = orc_max(likelihood, rfsize);
    *argmax = orc_maxidx(likelihood, rfsize);
    /* std::max_element: first maximum */
    double best = exp(log(likelihood[0]) + log(prior[0]));
    for (int j = 1; j < rfsize; j++) {
        double p = exp(log(likelihood[j]) + log(prior[j]));
        if (best < p) best = p;
    }
    *max_posterior = best;
}

static int size_of_factor(const orc_range *r)
{
    /* cafe/cafe_tree.c:46-67 */
    int rsize = r->root_max - r->root_min + 1;
    int fsize = r->max - r->min + 1;
    return rsize > fsize ? rsize : fsize;
}

double orc_eval_posterior(const orc_tree *t, int F, int n_leaves, const int *counts,
                          const int *ref, const orc_range *range,
                          const double *node_lambda, const double *node_mu,
                          const double *prior, const double *errormatrix, int err_mfs,
                          const unsigned char *leaf_has_err, int nthreads,
                          int *first_zero_family, double *max_lik, int *argmax_root,
                          double *max_post)
{
    /* reset_birthdeath_cache cafe/cafe_main.c:319-326 */
    int M = ORC_MAX(range->max, range->root_max);
    orc_matrices *mats = orc_matrices_build(t, node_lambda, node_mu, M, nthreads);
    int sof = size_of_factor(range);
    int rfsize = range->root_max - range->root_min + 1;
    double *fml = (double *)malloc(sizeof(double) * ORC_MAX(F, 1));
    double *fmp = (double *)malloc(sizeof(double) * ORC_MAX(F, 1));
    int *fam = (int *)malloc(sizeof(int) * ORC_MAX(F, 1));

    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double *L = (double *)malloc(sizeof(double) * (size_t)t->n_nodes * sof);
        double *lf = (double *)malloc(sizeof(double) * (size_t)sof * 2), *rf = lf + sof;
        int *fs = (int *)malloc(sizeof(int) * t->n_nodes);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int i = 0; i < F; i++) {
            /* get_posterior cafe/lambda.cpp:698-714: duplicates reuse the ref row's result */
            if (ref && ref[i] >= 0 && ref[i] != i) continue;
            for (int k = 0; k < t->n_nodes; k++) fs[k] = -1;
            for (int j = 0; j < n_leaves; j++) fs[2 * j] = counts[(size_t)i * n_leaves + j];
            tree_likelihoods_ws(t, range, mats, fs, errormatrix, err_mfs, leaf_has_err, L, sof, lf, rf);
            orc_compute_posterior(L + (size_t)t->root * sof, rfsize, prior, &fml[i], &fam[i],
                                  &fmp[i]);
        }
        free(L);
        free(lf);
        free(fs);
    }
    double score = 0;
    int fz = -1;
    for (int i = 0; i < F; i++) {
        if (ref && ref[i] >= 0 && ref[i] != i) {
            fml[i] = fml[ref[i]];
            fmp[i] = fmp[ref[i]];
            fam[i] = fam[ref[i]];
        }
        if (fml[i] == 0 && fz < 0) fz = i; /* cafe/lambda.cpp:715-720 throw -> score -inf */
        score += log(fmp[i]);              /* cafe/lambda.cpp:721 */
    }
    if (fz >= 0) score = log(0); /* cafe/lambda.cpp:753-760 */
    if (first_zero_family) *first_zero_family = fz;
    if (max_lik) memcpy(max_lik, fml, sizeof(double) * F);
    if (max_post) memcpy(max_post, fmp, sizeof(double) * F);
    if (argmax_root) memcpy(argmax_root, fam, sizeof(int) * F);
    free(fml);
    free(fmp);
    free(fam);
    orc_matrices_free(mats);
    return score;
}

void orc_eval_root_likelihoods(const orc_tree *t, int B, int n_leaves, const int *counts,
                               const int *root_lo, const int *root_hi, const int *col_max,
                               const orc_matrices *mats, double *out)
{
    int S = orc_matrices_size(mats);
    int sof = S + 1;
    for (int b = 0; b < B; b++) sof = ORC_MAX(sof, root_hi[b] - root_lo[b] + 1);
    double *L = (double *)malloc(sizeof(double) * (size_t)t->n_nodes * sof);
    int *fs = (int *)malloc(sizeof(int) * t->n_nodes);
    size_t off = 0;
    for (int b = 0; b < B; b++) {
        orc_range r;
        r.min = 0;
        r.max = col_max[b];
        r.root_min = root_lo[b];
        r.root_max = root_hi[b];
        for (int k = 0; k < t->n_nodes; k++) fs[k] = -1;
        for (int j = 0; j < n_leaves; j++) fs[2 * j] = counts[(size_t)b * n_leaves + j];
        orc_compute_tree_likelihoods(t, &r, mats, fs, NULL, 0, NULL, L, sof);
        int n = root_hi[b] - root_lo[b] + 1;
        memcpy(out + off, L + (size_t)t->root * sof, sizeof(double) * n);
        off += n;
    }
    free(L);
    free(fs);
}

/* ======================================================================== */
/* k-cluster posterior: cafe/cafe_main.c:165-253 over                          */
/* cafe_tree_clustered_likelihood cafe/cafe_tree.c:704-850                    */
/* ======================================================================== */

/* libtree/input_values.c:84-94 */
void orc_copy_weights(double *out, const double *parameters, int start, int count)
{
    int i;
    double sumofweights = 0;
    for (i = 0; i < count - 1; i++) {
        out[i] = parameters[start + i];
        sumofweights += parameters[start + i];
    }
    out[i] = 1 - sumofweights;
}

/* One evaluation of the clustered objective for K clusters.  Cluster k prunes with its own matrices
 * (node_lambda/node_mu + k*n_nodes): the per-node recursion of cafe_tree.c:789-812 is the dense product of
 * compute_internal_node_likelihood with k_bd[k] in place of the single matrix.  Per family
 * (cafe_main.c:180-211): MAP_k = max_j exp(log L_k[j] + log prior[j]) * weights[k] (the max runs over a zeroed
 * FAMILYSIZEMAX array, :190-196), p_z[k] = MAP_k / sum, MAP = sum_k p_z[k] * MAP_k; duplicates copy their ref row
 * (:220-229); the first family with MAP == 0 makes the score log(0) and stops the loop (:231-240); afterwards the
 * weights become the mean memberships (:243-245).  Returns the score. */
double orc_eval_clustered_posterior(const orc_tree *t, int F, int n_leaves, const int *counts, const int *ref,
                                    const orc_range *range, int K, const double *node_lambda,
                                    const double *node_mu, const double *weights, const double *prior,
                                    int nthreads, double *MAP_out, double *p_z_out, double *new_weights,
                                    int *first_zero_family)
{
    int M = ORC_MAX(range->max, range->root_max);
    int sof = size_of_factor(range);
    int rfsize = range->root_max - range->root_min + 1;
    orc_matrices **mats = (orc_matrices **)malloc(sizeof(*mats) * K);
    for (int k = 0; k < K; k++)
        mats[k] = orc_matrices_build(t, node_lambda + (size_t)k * t->n_nodes, node_mu + (size_t)k * t->n_nodes, M, nthreads);
    double *mp = (double *)malloc(sizeof(double) * (size_t)ORC_MAX(F, 1) * K);   /* max posterior per family, cluster */
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double *L = (double *)malloc(sizeof(double) * (size_t)t->n_nodes * sof);
        int *fs = (int *)malloc(sizeof(int) * t->n_nodes);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int i = 0; i < F; i++) {
            if (ref && ref[i] >= 0 && ref[i] != i) continue;
            for (int n = 0; n < t->n_nodes; n++) fs[n] = -1;
            for (int j = 0; j < n_leaves; j++) fs[2 * j] = counts[(size_t)i * n_leaves + j];
            for (int k = 0; k < K; k++) {
                orc_compute_tree_likelihoods(t, range, mats[k], fs, NULL, 0, NULL, L, sof);
                const double *lk = L + (size_t)t->root * sof;
                double best = 0; /* posterior[] is calloc'd: entries outside [root_min, root_min + rfsize) are 0 */
                for (int j = 0; j < rfsize; j++) {
                    double p = exp(log(lk[j]) + log(prior[j]));
                    if (best < p) best = p;
                }
                mp[(size_t)i * K + k] = best;
            }
        }
        free(L);
        free(fs);
    }
    double score = 0;
    int fz = -1;
    double *sumofweights = (double *)calloc(K, sizeof(double));
    double *MAP_k = (double *)malloc(sizeof(double) * K);
    double *pz = (double *)malloc(sizeof(double) * (size_t)ORC_MAX(F, 1) * K);
    double *MAP = (double *)malloc(sizeof(double) * ORC_MAX(F, 1));
    for (int i = 0; i < F; i++) {
        if (!(ref && ref[i] >= 0 && ref[i] != i)) {
            double sumLikelihood = 0;
            for (int k = 0; k < K; k++) {
                MAP_k[k] = mp[(size_t)i * K + k] * weights[k];
                sumLikelihood += MAP_k[k];
            }
            for (int k = 0; k < K; k++) {
                pz[(size_t)i * K + k] = MAP_k[k] / sumLikelihood;
                sumofweights[k] += pz[(size_t)i * K + k];
            }
            double expectedPosterior = 0;
            for (int k = 0; k < K; k++) expectedPosterior += pz[(size_t)i * K + k] * (MAP_k[k]);
            MAP[i] = expectedPosterior;
        } else {
            MAP[i] = MAP[ref[i]];
            for (int k = 0; k < K; k++) {
                pz[(size_t)i * K + k] = pz[(size_t)ref[i] * K + k];
                sumofweights[k] += pz[(size_t)i * K + k];
            }
        }
        if (MAP[i] == 0) {
            score = log(0);
            fz = i;
            break;
        }
        score += log(MAP[i]);
    }
    for (int k = 0; k < K; k++) new_weights[k] = sumofweights[k] / F;
    if (first_zero_family) *first_zero_family = fz;
    if (MAP_out) memcpy(MAP_out, MAP, sizeof(double) * F);
    if (p_z_out) memcpy(p_z_out, pz, sizeof(double) * (size_t)F * K);
    for (int k = 0; k < K; k++) orc_matrices_free(mats[k]);
    free(mats);
    free(mp);
    free(sumofweights);
    free(MAP_k);
    free(pz);
    free(MAP);
    return score;
}

/* ======================================================================== */
/* cafe/cafe_family.c                                                        */
/* ======================================================================== */

/* cafe/cafe_family.c:357-364 */
void orc_init_family_size(orc_range *fs, int max)
{
    fs->root_min = 1;
    fs->root_max = (int)ORC_MAX(30, rint(max * 1.25));
    fs->max = max + ORC_MAX(50, max / 5);
    fs->min = 0;
}

/* cafe/cafe_family.c:9-34 -- O(F^2) scan; ref = lowest index with the same row */
void orc_family_check_the_pattern(int F, int n, const int *counts, int *ref)
{
    for (int i = 0; i < F; i++) ref[i] = -1;
    for (int i = 0; i < F; i++) {
        if (ref[i] != -1) continue;
        ref[i] = i;
        for (int j = i + 1; j < F; j++) {
            if (ref[j] != -1) continue;
            int k;
            for (k = 0; k < n; k++) {
                if (counts[(size_t)i * n + k] != counts[(size_t)j * n + k]) break;
            }
            if (k == n) ref[j] = i;
        }
    }
}

/* ======================================================================== */
/* cafe/lambda.cpp: prior                                                    */
/* ======================================================================== */

/* cafe/lambda.cpp:841-852 */
void orc_prior_poisson(double *prior, int n, int shift, double lambda)
{
    for (int i = 0; i < n; i++) prior[i] = orc_poisspdf(shift - 1 + i, lambda);
}

/* cafe/lambda.cpp:771-806 */
double orc_lnLPoisson(double lambda, int F, int n, const int *counts)
{
    double score = 0;
    for (int idx = 0; idx < F; idx++) {
        for (int i = 0; i < n; i++) {
            int cnt = counts[(size_t)idx * n + i];
            if (cnt > 0) {
                double ll = orc_poisspdf(cnt - 1, lambda);
                if (isnan(ll)) ll = 0;
                score += log(ll);
            }
        }
    }
    return -score;
}

typedef struct {
    int F, n;
    const int *counts;
} poisson_args;

static double poisson_eq(double *x, void *a)
{
    poisson_args *pa = (poisson_args *)a;
    return orc_lnLPoisson(x[0], pa->F, pa->n, pa->counts);
}

/* cafe/lambda.cpp:808-838 (start value is unifrnd() there; passed in here) */
double orc_find_poisson_lambda(int F, int n, const int *counts, double x0, int *iters,
                               double *score)
{
    poisson_args pa = {F, n, counts};
    double xmin, fmin;
    int bymax;
    int it = orc_fminsearch(poisson_eq, 1, &pa, &x0, 1e-6, 1e-6, 10000, &xmin, &fmin, &bymax);
    if (iters) *iters = it;
    if (score) *score = fmin;
    return xmin;
}

/* ======================================================================== */
/* libcommon/fminsearch.cpp                                                  */
/* ======================================================================== */

typedef struct {
    int N, N1;
    double rho, chi, psi, sigma, tolx, tolf, delta, zero_delta;
    int maxiters;
    double **v, **vsort;
    double *fv, *x_mean, *x_r, *x_tmp;
    int *idx;
    orc_math_func eq;
    void *args;
} fms;

/* libcommon/fminsearch.cpp:77-107 */
static void qsort_double_with_index(double *list, int *idx, int left, int right)
{
    double pivot = list[left];
    int pivot_idx = idx[left];
    int from = left;
    int to = right;
    while (from < to) {
        while (pivot <= list[to] && from < to) to--;
        if (from != to) {
            list[from] = list[to];
            idx[from] = idx[to];
            from++;
        }
        while (pivot >= list[from] && from < to) from++;
        if (from != to) {
            list[to] = list[from];
            idx[to] = idx[from];
            to--;
        }
    }
    list[from] = pivot;
    idx[from] = pivot_idx;
    if (left < from) qsort_double_with_index(list, idx, left, from - 1);
    if (right > from) qsort_double_with_index(list, idx, from + 1, right);
}

/* :109-123 */
static void fms_sort(fms *p)
{
    for (int i = 0; i < p->N1; i++) p->idx[i] = i;
    qsort_double_with_index(p->fv, p->idx, 0, p->N);
    for (int i = 0; i < p->N1; i++) {
        int k = p->idx[i];
        for (int j = 0; j < p->N; j++) p->vsort[i][j] = p->v[k][j];
    }
    for (int i = 0; i < p->N1; i++) memcpy(p->v[i], p->vsort[i], sizeof(double) * p->N);
}

/* :126-141 */
static int fms_checkV(fms *p)
{
    double max = -DBL_MAX;
    for (int i = 0; i < p->N; i++) {
        for (int j = 0; j < p->N; j++) {
            double t = fabs(p->v[i + 1][j] - p->v[i][j]);
            if (t > max) max = t;
        }
    }
    return max <= p->tolx;
}

/* :143-154 */
static int fms_checkF(fms *p)
{
    double max = -DBL_MAX;
    for (int i = 1; i < p->N1; i++) {
        double t = fabs(p->fv[i] - p->fv[0]);
        if (t > max) max = t;
    }
    return max <= p->tolf;
}

/* :156-187 -- note the isinf(previous vertex value) rule */
static void fms_min_init(fms *p, const double *X0)
{
    for (int i = 0; i < p->N1; i++) {
        for (int j = 0; j < p->N; j++) {
            if (i > 1 && isinf(p->fv[i - 1])) {
                if ((i - 1) == j)
                    p->v[i][j] = X0[j] ? (1 + p->delta * 100) * X0[j] : p->zero_delta;
                else
                    p->v[i][j] = X0[j];
            } else {
                if ((i - 1) == j)
                    p->v[i][j] = X0[j] ? (1 + p->delta) * X0[j] : p->zero_delta;
                else
                    p->v[i][j] = X0[j];
            }
        }
        p->fv[i] = p->eq(p->v[i], p->args);
    }
    fms_sort(p);
}

static void fms_set_last(fms *p, const double *x, double f)
{
    /* :252-262 */
    for (int i = 0; i < p->N; i++) p->v[p->N][i] = x[i];
    p->fv[p->N] = f;
    fms_sort(p);
}

static void fms_shrink(fms *p)
{
    /* :238-250 */
    for (int i = 1; i < p->N1; i++) {
        for (int j = 0; j < p->N; j++)
            p->v[i][j] = p->v[0][j] + p->sigma * (p->v[i][j] - p->v[0][j]);
        p->fv[i] = p->eq(p->v[i], p->args);
    }
    fms_sort(p);
}

int orc_fminsearch(orc_math_func eq, int N, void *args, const double *x0, double tolx,
                   double tolf, int maxiters, double *xmin, double *fmin, int *bymax)
{
    fms P;
    fms *p = &P;
    /* fminsearch_new :7-21 */
    p->rho = 1;
    p->chi = 2;
    p->psi = 0.5;
    p->sigma = 0.5;
    p->tolx = tolx;
    p->tolf = tolf;
    p->delta = 0.05;
    p->zero_delta = 0.00025;
    p->maxiters = maxiters;
    p->N = N;
    p->N1 = N + 1;
    p->eq = eq;
    p->args = args;
    p->v = (double **)malloc(sizeof(double *) * p->N1);
    p->vsort = (double **)malloc(sizeof(double *) * p->N1);
    for (int i = 0; i < p->N1; i++) {
        p->v[i] = (double *)calloc(N, sizeof(double));
        p->vsort[i] = (double *)calloc(N, sizeof(double));
    }
    p->fv = (double *)calloc(p->N1, sizeof(double));
    p->x_mean = (double *)calloc(N, sizeof(double));
    p->x_r = (double *)calloc(N, sizeof(double));
    p->x_tmp = (double *)calloc(N, sizeof(double));
    p->idx = (int *)calloc(p->N1, sizeof(int));

    /* fminsearch_min :264-302 */
    int i;
    fms_min_init(p, x0);
    for (i = 0; i < p->maxiters; i++) {
        if (fms_checkV(p) && fms_checkF(p)) break;
        /* x_mean :189-201 */
        for (int a = 0; a < N; a++) {
            p->x_mean[a] = 0;
            for (int j = 0; j < N; j++) p->x_mean[a] += p->v[j][a];
            p->x_mean[a] /= N;
        }
        /* reflection :203-211 */
        for (int a = 0; a < N; a++)
            p->x_r[a] = p->x_mean[a] + p->rho * (p->x_mean[a] - p->v[N][a]);
        double fv_r = p->eq(p->x_r, p->args);
        if (fv_r < p->fv[0]) {
            /* expansion :214-222 */
            for (int a = 0; a < N; a++)
                p->x_tmp[a] = p->x_mean[a] + p->chi * (p->x_r[a] - p->x_mean[a]);
            double fv_e = p->eq(p->x_tmp, p->args);
            if (fv_e < fv_r)
                fms_set_last(p, p->x_tmp, fv_e);
            else
                fms_set_last(p, p->x_r, fv_r);
        } else if (fv_r >= p->fv[N]) {
            if (fv_r > p->fv[N]) {
                /* contract inside :233-241 */
                for (int a = 0; a < N; a++)
                    p->x_tmp[a] = p->x_mean[a] + p->psi * (p->x_mean[a] - p->v[N][a]);
                double fv_cc = p->eq(p->x_tmp, p->args);
                if (fv_cc < p->fv[N])
                    fms_set_last(p, p->x_tmp, fv_cc);
                else
                    fms_shrink(p);
            } else {
                /* contract outside :224-231 */
                for (int a = 0; a < N; a++)
                    p->x_tmp[a] = p->x_mean[a] + p->psi * (p->x_r[a] - p->x_mean[a]);
                double fv_c = p->eq(p->x_tmp, p->args);
                if (fv_c <= fv_r)
                    fms_set_last(p, p->x_tmp, fv_c);
                else
                    fms_shrink(p);
            }
        } else {
            fms_set_last(p, p->x_r, fv_r);
        }
    }
    if (bymax) *bymax = (i == p->maxiters);
    for (int a = 0; a < N; a++) xmin[a] = p->v[0][a];
    if (fmin) *fmin = p->fv[0];
    for (int k = 0; k < p->N1; k++) {
        free(p->v[k]);
        free(p->vsort[k]);
    }
    free(p->v);
    free(p->vsort);
    free(p->fv);
    free(p->x_mean);
    free(p->x_r);
    free(p->x_tmp);
    free(p->idx);
    return i;
}

/* ======================================================================== */
/* objective: cafe/lambda.cpp:726-769, cafe/lambdamu.cpp:323-367             */
/* ======================================================================== */

double orc_lambda_objective(double *x, void *args)
{
    orc_objective *o = (orc_objective *)args;
    int num_params = o->num_lambdas * (o->has_mu ? 2 : 1);
    double score = 0;
    int skip = 0;
    /* cafe/lambda.cpp:733-741 tests x[0..num_lambdas); cafe/lambdamu.cpp:331-339 tests all */
    for (int i = 0; i < (o->has_mu ? num_params : o->num_lambdas); i++) {
        if (x[i] < 0) {
            skip = 1;
            score = log(0);
            break;
        }
    }
    if (!skip) {
        int n = o->t->n_nodes;
        double *nl = (double *)malloc(sizeof(double) * n);
        double *nm = (double *)malloc(sizeof(double) * n);
        for (int i = 0; i < n; i++) {
            int cls = o->node_class ? o->node_class[i] : 0;
            if (cls < 0) cls = 0; /* cafe/cafe_shell.c:150-156 */
            nl[i] = x[cls];
            nm[i] = o->has_mu ? x[o->num_lambdas + cls] : -1; /* cafe/cafe_shell.c:172-175, 138-141 */
        }
        score = orc_eval_posterior(o->t, o->F, o->n_leaves, o->counts, o->ref, &o->range, nl, nm,
                                   o->prior, NULL, 0, NULL, o->nthreads, NULL, NULL, NULL, NULL);
        free(nl);
        free(nm);
    }
    if (o->trace && o->n_evals < o->trace_cap) {
        double *row = o->trace + (size_t)o->n_evals * (num_params + 1);
        for (int i = 0; i < num_params; i++) row[i] = x[i];
        row[num_params] = score;
    }
    o->n_evals++;
    return -score;
}

/* ======================================================================== */
/* Monte-Carlo null                                                          */
/* ======================================================================== */

static void random_familysize_rec(const orc_tree *t, const orc_matrices *mats, int node,
                                  int max_family_size, int *familysize, int *max)
{
    /* tree_traveral_prefix (libtree/tree.c:101-124): node, then left subtree, then right.
     * __cafe_tree_node_random_familysize cafe/cafe_tree.c:533-561 */
    if (node != t->root) {
        double rnd = orc_unifrnd();
        double cumul = 0;
        int S = orc_matrices_size(mats);
        const double *m = orc_matrices_get(mats, node);
        int parent_size = familysize[t->parent[node]];
        int c = 0;
        for (; c < max_family_size - 1; c++) {
            cumul += m[(size_t)parent_size * S + c];
            if (cumul >= rnd) break;
        }
        familysize[node] = c;
        if (*max < c) *max = c;
    }
    if (t->left[node] >= 0) {
        random_familysize_rec(t, mats, t->left[node], max_family_size, familysize, max);
        random_familysize_rec(t, mats, t->right[node], max_family_size, familysize, max);
    }
}

/* cafe/cafe_tree.c:563-569 */
int orc_tree_random_familysize(const orc_tree *t, const orc_matrices *mats, int root_size,
                               int max_family_size, int *familysize)
{
    int max = 0;
    familysize[t->root] = root_size;
    random_familysize_rec(t, mats, t->root, max_family_size, familysize, &max);
    return max;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* cafe/conditional_distribution.cpp:10-57, single thread order (-t 1) */
void orc_conditional_distribution(const orc_tree *t, const orc_range *range,
                                  const orc_matrices *mats, int trials, double *out)
{
    int R = range->root_max - range->root_min + 1;
    int sof = ORC_MAX(orc_matrices_size(mats) + 1, R);
    double *L = (double *)malloc(sizeof(double) * (size_t)t->n_nodes * sof);
    int *fs = (int *)malloc(sizeof(int) * t->n_nodes);
    for (int k = 0; k < t->n_nodes; k++) fs[k] = -1;
    for (int s = range->root_min; s <= range->root_max; s++) {
        /* get_random_probabilities :10-44 */
        orc_range r = *range;
        r.root_min = s;
        r.root_max = s;
        int maxFamilySize = ORC_MAX(r.root_max, r.max);
        double *probs = out + (size_t)(s - range->root_min) * trials;
        for (int i = 0; i < trials; i++) {
            int max = orc_tree_random_familysize(t, mats, s, maxFamilySize, fs);
            r.max = ORC_MIN(max + ORC_MAX(50, max / 5), r.max); /* :29 running MIN */
            orc_compute_tree_likelihoods(t, &r, mats, fs, NULL, 0, NULL, L, sof);
            probs[i] = L[(size_t)t->root * sof + 0];
        }
        qsort(probs, trials, sizeof(double), cmp_double);
    }
    free(L);
    free(fs);
}

/* ======================================================================== */
/* Viterbi / p-values                                                        */
/* ======================================================================== */

/* cafe/cafe_family.c:236-255 */
void orc_family_forced_range(orc_range *r, int n_leaves, const int *row)
{
    int max = 0;
    for (int i = 0; i < n_leaves; i++)
        if (max < row[i]) max = row[i];
    r->min = 0;
    r->root_min = 1;
    r->root_max = (int)rint(max * 1.25);
    r->max = max + ORC_MAX(50, max / 5);
}

static void viterbi_compute(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                            const int *familysize, int *vit, double *L, int sof, int node,
                            double *f0, double *f1)
{
    /* __cafe_tree_node_compute_viterbi cafe/viterbi.cpp:208-320, post-order */
    double *Lv = L + (size_t)node * sof;
    if (t->left[node] < 0) {
        memset(Lv, 0, sizeof(double) * sof);
        if (familysize[node] >= 0 && familysize[node] < sof) Lv[familysize[node]] = 1; /* :262-266 */
        return;
    }
    viterbi_compute(t, range, mats, familysize, vit, L, sof, t->left[node], f0, f1);
    viterbi_compute(t, range, mats, familysize, vit, L, sof, t->right[node], f0, f1);
    int row_lo, row_hi;
    if (node == t->root) {
        row_lo = range->root_min;
        row_hi = range->root_max;
    } else {
        row_lo = range->min;
        row_hi = range->max;
    }
    int S = orc_matrices_size(mats);
    int child[2] = {t->left[node], t->right[node]};
    double *fac[2] = {f0, f1};
    for (int idx = 0; idx < 2; idx++) {
        const double *m = orc_matrices_get(mats, child[idx]);
        const double *Lc = L + (size_t)child[idx] * sof;
        int *vc = vit + (size_t)child[idx] * sof;
        memset(fac[idx], 0, sizeof(double) * sof);
        for (int s = row_lo, i = 0; s <= row_hi; s++, i++) {
            for (int c = range->min, j = 0; c <= range->max; c++, j++) {
                double tmp = m[(size_t)s * S + c] * Lc[j];
                if (tmp > fac[idx][i]) { /* strict: first maximum wins :296-300 */
                    fac[idx][i] = tmp;
                    vc[i] = j;
                }
            }
        }
    }
    int size = row_hi - row_lo + 1;
    for (int i = 0; i < size; i++) Lv[i] = f0[i] * f1[i];
}

static void viterbi_backtrack(const orc_tree *t, const orc_range *range, int *familysize,
                              const int *vit, const double *L, int sof, int node)
{
    /* __cafe_tree_node_backtrack_viterbi cafe/viterbi.cpp:322-351, prefix order */
    if (!(t->left[node] < 0 && familysize[node] >= 0)) {
        if (node == t->root) {
            int rfsize = range->root_max - range->root_min + 1;
            familysize[node] = range->root_min + (rfsize > 0 ? orc_maxidx(L + (size_t)node * sof, rfsize) : 0);
        } else {
            int parent = t->parent[node];
            int base = (parent == t->root) ? range->root_min : range->min;
            familysize[node] = vit[(size_t)node * sof + (familysize[parent] - base)] + range->min;
        }
    }
    if (t->left[node] >= 0) {
        viterbi_backtrack(t, range, familysize, vit, L, sof, t->left[node]);
        viterbi_backtrack(t, range, familysize, vit, L, sof, t->right[node]);
    }
}

void orc_tree_viterbi(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                      int *familysize, int *vit, double *L, int sof)
{
    double *f0 = (double *)calloc(sof, sizeof(double));
    double *f1 = (double *)calloc(sof, sizeof(double));
    /* internal nodes start unknown */
    for (int i = 0; i < t->n_nodes; i++)
        if (t->left[i] >= 0) familysize[i] = -1;
    viterbi_compute(t, range, mats, familysize, vit, L, sof, t->root, f0, f1);
    viterbi_backtrack(t, range, familysize, vit, L, sof, t->root);
    free(f0);
    free(f1);
}

void orc_tree_p_values(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                       const int *familysize, const double *cd, int trials, double *pvalues)
{
    int rfsize = range->root_max - range->root_min + 1;
    int sof = ORC_MAX(orc_matrices_size(mats) + 1, ORC_MAX(rfsize, 1));
    double *L = (double *)malloc(sizeof(double) * (size_t)t->n_nodes * sof);
    orc_compute_tree_likelihoods(t, range, mats, familysize, NULL, 0, NULL, L, sof);
    const double *obs = L + (size_t)t->root * sof;
    for (int s = 0; s < rfsize; s++) pvalues[s] = orc_pvalue(obs[s], cd + (size_t)s * trials, trials);
    free(L);
}

void orc_viterbi_sum_probabilities(const orc_tree *t, const orc_range *range, const orc_matrices *mats,
                                   const int *familysize, double *out)
{
    int S = orc_matrices_size(mats);
    int nnodes = (t->n_nodes - 1) / 2;
    for (int j = 0; j < nnodes; j++) {
        int node = 2 * j + 1;
        int child[2] = {t->left[node], t->right[node]};
        for (int k = 0; k < 2; k++) {
            const double *m = orc_matrices_get(mats, child[k]);
            double p = m[(size_t)familysize[node] * S + familysize[child[k]]];
            double acc = 0;
            for (int mm = 0; mm <= range->max; mm++) {
                double v = m[(size_t)familysize[node] * S + mm];
                if (v == p)
                    acc += v / 2.0;
                else if (v < p)
                    acc += v;
            }
            out[2 * j + k] = acc;
        }
    }
}
