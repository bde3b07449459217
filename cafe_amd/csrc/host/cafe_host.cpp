// cafe_host.cpp -- host-side driver above the kernel boundary (include/cafehost.h).
//
// CAFE's command language for the hot path, restated in C++17: tree / family-table
// ingest, the Poisson root prior, the parameter scatter, the Nelder-Mead search and the
// log lines of `lambda` and `lambdamu`.  Every objective evaluation is ONE call of
// cafehip_eval_posterior (include/cafehip.h) on the GPU; nothing here computes a
// likelihood.  Each function cites the reference behaviour it reproduces (file:line).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <dirent.h>
#include <dlfcn.h>
#include <thread>
#include <atomic>
#include <map>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../../include/cafehip.h"
#include "../../../include/cafehost.h"
#include "../host_math.hpp"
#include "glibc_rand.hpp"
#include "host_util.hpp"
#include "nelder_mead.hpp"
#include "poisson_prior.hpp"
#include "tree_table.hpp"

namespace {

thread_local std::string g_host_err;

int host_fail(const std::string& msg)
{
    g_host_err = msg;
    return -1;
}

using namespace cafehost_impl;

}  // namespace

// ====================================================================================
// session == CafeParam
// ====================================================================================
// Inverse-CDF sampling of a child size given the parent size (cafe_tree_random_familysize,
// cafe/cafe_tree.c:533-569): the reference accumulates the matrix row entry by entry until the running sum
// reaches the random number.  The running sums of a row do not depend on the draw, so they are formed once
// per (node, parent size) -- by the same sequence of additions -- and each draw is a binary search for the
// first prefix >= rnd (prefixes of non-negative terms are non-decreasing), capped like the reference's loop.
struct CdfCache {
    const std::vector<std::vector<double>>* mats = nullptr;
    int S = 0;
    std::vector<std::vector<double>> rows;  // [node * S + parent_size], empty until first use
    void reset(const std::vector<std::vector<double>>& m, int side)
    {
        mats = &m;
        S = side;
        rows.assign(m.size() * (size_t)side, {});
    }
    void fill(int node, int parent_size)
    {
        std::vector<double>& pre = rows[(size_t)node * S + parent_size];
        pre.resize(S);
        const double* m = (*mats)[node].data() + (size_t)parent_size * S;
        double cumul = 0;
        for (int c = 0; c < S; ++c) {
            cumul += m[c];
            pre[c] = cumul;
        }
    }
    void build_all()  // every row now, so that concurrent draws only read
    {
        for (size_t node = 0; node < mats->size(); ++node)
            if (!(*mats)[node].empty())
                for (int ps = 0; ps < S; ++ps) fill((int)node, ps);
    }
    int draw(int node, int parent_size, double rnd, int max_family_size)
    {
        std::vector<double>& pre = rows[(size_t)node * S + parent_size];
        if (pre.empty()) fill(node, parent_size);
        const int limit = std::min(std::max(max_family_size - 1, 0), S);
        return (int)(std::lower_bound(pre.begin(), pre.begin() + limit, rnd) - pre.begin());
    }
};

struct cafehost_session {
    cafehip_ctx* ctx = nullptr;
    cafehip_ctx* ctx_one = nullptr;  // one-family evaluations (lambda -e), created on first use
    int device_id = 0;
    GlibcRand rng;
    double unifrnd() { return rng.unifrnd(); }
    FILE* flog = stdout;
    bool own_log = false;
    std::string log_name = "stdout";
    bool quiet = false;

    bool have_tree = false, have_family = false;
    HostTree tree;
    HostTree lambda_tree;
    bool have_lambda_tree = false;
    std::vector<int> node_class;  // taxaid per node (cafe/cafe_shell.c:324-332), -1.. -> 0
    HostFamilies fam;
    std::vector<int> species_index;  // species -> node id, -1 if absent (cafe/gene_family.cpp:413-445)
    HostRange range;
    double pvalue = 0.01;
    int num_threads = 1;
    int num_random_samples = 1000;
    bool device_families_current = false;
    // multi-GPU: this rank's chunk-aligned block [shard_lo, shard_hi) of the table
    int shard_rank = 0, shard_world = 1, shard_lo = 0, shard_hi = 0;
    cafehost_exchange_fn exchange = nullptr;
    void* exchange_user = nullptr;
    double* d_exch_chunks = nullptr;
    int32_t* d_exch_fz = nullptr;
    cafehost_allgather_fn allgather = nullptr;  // report / Monte-Carlo null sharding
    void* allgather_user = nullptr;
    // native communicator (cafehost_init_comm): the exchange lives behind the kernel library's ABI
    // (cafehip_comm_*, include/cafehip.h); this session only tells it every rank's block of the table
    bool native_comm = false;
    std::vector<int32_t> all_lo, all_hi;                  // every rank's block of the table
    // an UNSHARDED session inside a sharded job: while set, this rank loads whole tables and evaluates them alone on
    // its GPU (lhtest deals its files to the ranks: whole searches per GPU instead of 59-family tables cut N ways)
    bool solo = false;
    int opt_lhtest_deal = 1;                              // cafehost_set_option "lhtest_deal": 0 = every rank runs every file, sharded
    int opt_grid_deal = -1;                               // "grid_deal": lambda -r deals its grid points to the ranks: -1 where the table does not fill a GPU, 0 never, 1 always
    long long grid_points_dealt = 0;
    bool mute_log = false;

    void hipapi_check(hipError_t e, const char* what)
    {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }

    // Every rank knows every rank's block (same rule, same F), so nothing is communicated to size the exchange.
    // Called by upload(): a `load` or `tree` re-wires automatically.
    void wire_native()
    {
        const int Fall = fam.F();
        const int n_chunks = (Fall + CAFEHIP_CHUNK - 1) / CAFEHIP_CHUNK;
        const int base = n_chunks / shard_world, extra = n_chunks % shard_world;
        all_lo.assign(shard_world, 0);
        all_hi.assign(shard_world, 0);
        int c0 = 0;
        for (int r = 0; r < shard_world; ++r) {
            const int nc = base + (r < extra ? 1 : 0);
            all_lo[r] = std::min(c0 * CAFEHIP_CHUNK, Fall);
            all_hi[r] = std::min((c0 + nc) * CAFEHIP_CHUNK, Fall);
            c0 += nc;
        }
        hip_check(cafehip_comm_set_blocks(ctx, all_lo.data(), all_hi.data()));
        allgather = &cafehost_session::native_allgather;
        allgather_user = this;
    }

    // report / Monte-Carlo null: fixed-slot all-gather of host blocks
    static int native_allgather(void* user, const void* mine, long long nbytes_mine, void* all, long long nbytes_slot)
    {
        cafehost_session* s = static_cast<cafehost_session*>(user);
        if (cafehip_comm_allgather(s->ctx, mine, (size_t)nbytes_mine, all, (size_t)nbytes_slot) != 0) {
            fprintf(stderr, "allgather failed: %s\n", cafehip_last_error());
            return -1;
        }
        return 0;
    }

    bool report_sharded() const { return shard_world > 1 && allgather != nullptr; }
    // report-phase commands read `params` as ONE model's rates (node_rates); after `lambda -k` it holds K x lambdas
    // plus K - 1 weights
    void require_single_model(const char* what) const
    {
        if (k_clusters > 0)
            throw std::runtime_error(std::string(what) + " is not supported after `lambda -k` (clustered model): set a single model with lambda / lambdamu first");
    }
    // contiguous block of [0, n) owned by rank r
    void block_of(int n, int r, int& lo, int& hi) const
    {
        const int base = n / shard_world, extra = n % shard_world;
        lo = r * base + std::min(r, extra);
        hi = lo + base + (r < extra ? 1 : 0);
    }
    // Every rank contributes `bytes_of(rank)` bytes (sizes are a function of the shared table, so every rank
    // knows them all); returns the blocks concatenated in rank order.  One collective.
    template <class SizeFn>
    std::vector<char> gather_blocks(const void* mine, SizeFn bytes_of)
    {
        long long slot = 0, total = 0;
        for (int r = 0; r < shard_world; ++r) {
            slot = std::max<long long>(slot, bytes_of(r));
            total += bytes_of(r);
        }
        std::vector<char> all((size_t)std::max<long long>(slot, 1) * shard_world), out((size_t)total);
        if (allgather(allgather_user, mine, bytes_of(shard_rank), all.data(), std::max<long long>(slot, 1)) != 0)
            throw std::runtime_error("allgather callback failed");
        size_t o = 0;
        for (int r = 0; r < shard_world; ++r) {
            memcpy(out.data() + o, all.data() + (size_t)r * std::max<long long>(slot, 1), (size_t)bytes_of(r));
            o += (size_t)bytes_of(r);
        }
        return out;
    }

    // model state
    int num_lambdas = 1, num_mus = 0, num_params = 0;
    bool has_mu = false, eqbg = false, checkconv = false;
    std::vector<double> params;
    std::vector<double> prior;  // FAMILYSIZEMAX entries
    double poisson_lambda = 0, last_score = 0;
    int search_iters = 0, n_evals = 0;
    double search_seconds = 0;
    std::vector<double> trace;
    // k-cluster model (`lambda -k`, cafe/lambda.cpp:273-352): K clusters, each with its own rates, mixed by weights
    int k_clusters = 0;
    bool fixcluster0 = false;                 // -f: cluster 0 has lambda fixed at 0
    std::vector<double> k_weights;            // param->k_weights: mean memberships after every evaluation
    std::vector<double> last_membership_sums; // sum over families of p_z[k] of the LAST evaluation
    std::vector<double> p_z_membership;       // F x K of the last evaluation (kept for the membership log)
    std::vector<std::vector<double>> cond_dist;  // ConditionalDistribution::matrix, cafe/pvalue.cpp:13
    std::vector<int> root_dist;  // param->root_dist: families per root size (index = size), cafe_commands.cpp:742
    // error model (one model file; ErrorStruct, libtree/family.h:31-38)
    std::string err_file;
    int err_mfs = -1, err_fromdiff = 0, err_todiff = 0;
    std::vector<double> err_matrix;        // (mfs+1)^2, [observed][true]
    std::vector<uint8_t> err_leaf;         // per node: leaf carries the model
    // last report (exposed for tests)
    std::vector<double> rep_max_p;
    std::vector<int32_t> rep_sizes;      // F x n_nodes
    std::vector<double> rep_branch_p;    // F x (n_nodes-1), -1 = not computed
    std::vector<double> rep_avg_exp;     // per (node, child) pair: mean size change (compute_size_deltas)
    std::vector<int> rep_expand, rep_remain, rep_decrease;

    // phylogeny_string (libtree/phylogeny.c:490-540): names + ":%g" branch lengths; `label` adds a suffix
    std::string tree_string(const std::function<std::string(int)>& label, bool with_bl) const
    {
        std::function<std::string(int)> rec = [&](int v) -> std::string {
            std::string out;
            if (tree.left[v] >= 0) out = "(" + rec(tree.left[v]) + "," + rec(tree.right[v]) + ")";
            out += label(v);
            if (with_bl && tree.bl[v] >= 0) {
                char buf[64];
                snprintf(buf, sizeof buf, ":%g", tree.bl[v]);
                out += buf;
            }
            return out;
        };
        return rec(tree.root);
    }

    void log(const char* fmt, ...)
    {  // cafe_log, cafe/cafe_main.c:26-44
        if (mute_log) return;
        va_list ap;
        va_start(ap, fmt);
        vfprintf(flog, fmt, ap);
        va_end(ap);
        fflush(flog);
    }

    void hip_check(int rc)
    {
        if (rc != 0) throw std::runtime_error(std::string("cafehip: ") + cafehip_last_error());
    }

    // cafe_family_set_species_index, cafe/gene_family.cpp:413-445
    void sync_species_index()
    {
        if (!have_tree || !have_family) return;
        species_index.assign(fam.species.size(), -1);
        std::vector<bool> leaf_found(tree.n, false);
        for (size_t s = 0; s < fam.species.size(); ++s) {
            for (int i = 0; i < tree.n; i += 2) {
                if (iequals(fam.species[s], tree.name[i])) {
                    species_index[s] = i;
                    leaf_found[i] = true;
                    break;
                }
            }
        }
        for (int i = 0; i < tree.n; i += 2)
            if (!leaf_found[i]) throw std::runtime_error("No species '" + tree.name[i] + "' was found in the tree");
        device_families_current = false;
    }

    void upload()
    {
        if (device_families_current) return;
        spec.clear();   // cached scores belong to the table / tree / error model that was on the device
        hip_check(cafehip_set_tree(ctx, tree.n, tree.parent.data(), tree.left.data(), tree.right.data(), tree.bl.data()));
        const int nl = tree.n_leaves();
        const int Fall = fam.F();
        const int ns = (int)fam.species.size();
        {
            // contiguous chunk-aligned blocks (same rule as cafe_amd/distributed.py shard_bounds)
            const int n_chunks = (Fall + CAFEHIP_CHUNK - 1) / CAFEHIP_CHUNK;
            const int world_ = solo ? 1 : shard_world, rank_ = solo ? 0 : shard_rank;   // solo: the whole table is mine
            const int base = n_chunks / world_, extra = n_chunks % world_;
            int c0 = 0;
            for (int r = 0; r < rank_; ++r) c0 += base + (r < extra ? 1 : 0);
            const int nc = base + (rank_ < extra ? 1 : 0);
            shard_lo = std::min(c0 * CAFEHIP_CHUNK, Fall);
            shard_hi = std::min((c0 + nc) * CAFEHIP_CHUNK, Fall);
        }
        const int F = shard_hi - shard_lo;
        std::vector<int32_t> counts((size_t)std::max(F, 1) * nl, 0);
        for (int i = 0; i < F; ++i)
            for (int s = 0; s < ns; ++s)
                if (species_index[s] >= 0)
                    counts[(size_t)i * nl + species_index[s] / 2] = fam.counts[(size_t)(shard_lo + i) * ns + s];
        hip_check(cafehip_set_families(ctx, F, nl, counts.data(), nullptr, range.min, range.max, range.root_min, range.root_max));
        if (opt_objective_reference) device_rows = counts;   // (the reference-arithmetic objective scores the rows one by one)
        else device_rows.clear();
        if (err_mfs >= 0)
            hip_check(cafehip_set_error_model(ctx, err_mfs, err_matrix.data(), err_leaf.data()));
        else
            hip_check(cafehip_set_error_model(ctx, 0, nullptr, nullptr));
        device_families_current = true;
        if (native_comm && !solo) wire_native();
    }

    // The Poisson fit of the prior (collect_leaf_sizes + find_poisson_lambda, cafe/lambda.cpp:771-838) needs the table and one
    // random number; the table's upload (row dedup, compression plan, device copies) needs neither the fit nor the random
    // stream: a search command starts the fit on a host thread, uploads, and picks the fit up where the reference runs it
    // (round 5: configs[1] `lambda -s` spent 6.8 of 14 ms in the two, one after the other).  The random start is drawn on
    // the calling thread, in the reference's order; log lines are written when the result is picked up.
    struct PriorJob {
        bool started = false;
        std::thread worker;
        std::exception_ptr error;
        PoissonFit fit;
        size_t n_leaf_sizes = 0;
    } prior_job;
    void begin_prior_fit()
    {
        if (prior_job.started || !opt_prior_file.empty()) return;
        prior_job.error = nullptr;
        prior_job.fit = PoissonFit();
        const double start = unifrnd();
        const bool look = opt_prior_lookahead != 0;
        auto body = [this, start, look] {
            try {
                std::vector<int> leaf_sizes;  // collect_leaf_sizes :789-806
                const int ns = (int)fam.species.size();
                for (int idx = 0; idx < fam.F(); ++idx)
                    for (int i = 0; i < ns; ++i) {
                        if (species_index[i] < 0) continue;
                        const int cnt = fam.counts[(size_t)idx * ns + i];
                        if (cnt > 0) leaf_sizes.push_back(cnt - 1);
                    }
                prior_job.n_leaf_sizes = leaf_sizes.size();
                prior_job.fit.run(leaf_sizes, start, look);
            } catch (...) {
                prior_job.error = std::current_exception();
            }
        };
        // (started only once there is a worker -- or a finished fit: a thread that cannot be created runs the fit inline)
        try {
            prior_job.worker = std::thread(body);
        } catch (const std::system_error&) {
            body();
        }
        prior_job.started = true;
    }
    // a command that failed between starting the fit and picking it up must not leave it to the next command
    void drop_prior_job()
    {
        if (prior_job.worker.joinable()) prior_job.worker.join();
        prior_job.started = false;
        prior_job.error = nullptr;
    }
    // upload() with the prior fit running beside it; the caller calls set_prior_rfsize_empirical() where the reference does
    void upload_beside_prior_fit()
    {
        begin_prior_fit();
        try {
            upload();
        } catch (...) {
            drop_prior_job();
            throw;
        }
    }

    // ---- prior: cafe_set_prior_rfsize_empirical, cafe/lambda.cpp:808-870 ----
    void set_prior_rfsize_empirical()
    {
        spec.clear();   // ... and to the prior they were computed under
        if (!opt_prior_file.empty()) {
            // Not in the reference (its searches always fit the Poisson below): a root-size prior given by the caller,
            // one probability per line for root sizes root_min, root_min + 1, ... -- e.g. the distribution a simulated
            // table was drawn from (bench.py's `generator_prior` search leg)
            std::ifstream in(opt_prior_file);
            if (!in) throw std::runtime_error("ERROR(prior_file): Cannot open " + opt_prior_file + " in read mode.");
            prior.assign(1000, 0.0);
            double v, total = 0;
            int n = 0;
            while (n < 1000 && (in >> v)) {
                if (!(v >= 0) || !std::isfinite(v))
                    throw std::runtime_error("ERROR(prior_file): value " + std::to_string(n + 1) + " of " + opt_prior_file + " is not a finite probability >= 0");
                total += v;
                prior[n++] = v;
            }
            if (n == 0) throw std::runtime_error("ERROR(prior_file): no values in " + opt_prior_file);
            const int need = range.root_max - range.root_min + 1;
            if (n < need)
                throw std::runtime_error("ERROR(prior_file): " + opt_prior_file + " holds " + std::to_string(n) + " values but the root sizes " +
                                         std::to_string(range.root_min) + ".." + std::to_string(range.root_max) + " need " + std::to_string(need));
            if (std::fabs(total - 1.0) > 1e-6 && shard_rank == 0)
                fprintf(stderr, "WARNING(prior_file): the %d values of %s sum to %.9g, not 1\n", n, opt_prior_file.c_str(), total);
            log("Root size prior read from %s (%d values)\n", opt_prior_file.c_str(), n);
            (void)unifrnd();   // the fit's random start: the stream stays where the reference's flow would leave it
            return;
        }
        // (the fit proper may already be running beside the table's upload: begin_prior_fit)
        if (!prior_job.started) begin_prior_fit();
        if (prior_job.worker.joinable()) prior_job.worker.join();
        prior_job.started = false;
        if (prior_job.error) {
            std::exception_ptr e = prior_job.error;
            prior_job.error = nullptr;
            std::rethrow_exception(e);
        }
        PoissonFit& fit = prior_job.fit;
        poisson_lambda = fit.lambda;
        if (opt_timing && shard_rank == 0)
            fprintf(stderr, "prior fit: %zu leaf sizes, %d iterations, %ld passes of %ld chains, %ld cached / %ld single evaluations\n", prior_job.n_leaf_sizes,
                    fit.iters, fit.passes, fit.chains, fit.hits, fit.misses);
        log("Empirical Prior Estimation Result: (%d iterations)\n", fit.iters);
        log("Poisson lambda: %f & Score: %f\n", poisson_lambda, fit.score);
        prior.assign(1000, 0.0);  // cafe_set_prior_rfsize_poisson_lambda :841-852
        for (int i = 0; i < 1000; ++i) prior[i] = poisspdf(range.root_min - 1 + i, poisson_lambda);
    }

    // ---- cafe_shell_set_lambdas (cafe/cafe_shell.c:31-38) -> per-node (lambda, mu) ----
    void node_rates(const double* x, std::vector<double>& nl, std::vector<double>& nm) const
    {
        nl.assign(tree.n, 0.0);
        nm.assign(tree.n, -1.0);
        for (int i = 0; i < tree.n; ++i) {
            int cls = have_lambda_tree ? node_class[i] : 0;
            if (cls < 0) cls = 0;  // cafe/cafe_shell.c:150-156
            if (!has_mu) {
                nl[i] = x[cls];  // set_birth_death_probabilities4 :172-175
                nm[i] = -1;
            } else if (!have_lambda_tree) {
                nl[i] = x[0];  // set_birth_death_probabilities :48-52
                nm[i] = x[num_lambdas];
            } else if (eqbg) {
                nl[i] = x[cls];  // set_birth_death_probabilities2 :127-136
                nm[i] = (cls == 0) ? nl[i] : x[num_lambdas + (cls - 1)];
            } else {
                nl[i] = x[cls];  // :138-141
                nm[i] = x[num_lambdas + cls];
            }
        }
    }

    // ---- speculative / batched evaluation ----------------------------------------------
    // A table that fills less than half the chip leaves most compute units idle during an evaluation, and every
    // evaluation pays the same launch and walk latency.  The points Nelder-Mead may ask for next are known before
    // it asks (FMinSearch::prefetch), so they are evaluated TOGETHER (cafehip_eval_posterior_multi: one matrix
    // launch, one pruning launch with a set dimension) and objective() takes its value from this cache.  The values
    // are bit-identical to single evaluations, so the trajectory, the fitted parameters, the iteration count and the
    // log lines are those of the sequential run.  CAFEHOST_SPECULATE=0/1 forces it off/on.
    struct SpecEntry {
        std::vector<double> x;
        double score;
        int32_t zero;
    };
    std::vector<SpecEntry> spec;
    long spec_launches = 0, spec_points = 0, spec_hits = 0;
    int opt_speculate = -1;      // cafehost_set_option "speculate": -1 by how full the chip is, 0 off, 1 on
    bool opt_timing = false;     // "timing": phase times of report / the Monte-Carlo null on stderr
    std::string opt_prior_file;  // "prior_file": root-size prior of the searches read from a file instead of fitted
    int opt_prior_lookahead = 1; // "prior_lookahead": the Poisson fit evaluates the points Nelder-Mead may ask for several per pass
    // "report_arith" = reference: the likelihood vectors of the report phase (Monte-Carlo null, observed families) in the
    // reference's own arithmetic -- the row-per-thread kernel with a separate multiply and add per term (cafehip option
    // k2=v1ref) on the exact-form matrices: bit for bit what the reference's build computes on this machine, so every
    // likelihood's rank in its sorted null is the reference's for any input; slower than the matrix cores (default: fast)
    bool opt_report_reference = false;
    // "objective_arith" = reference: every objective evaluation of a search in the reference's own arithmetic AND order --
    // exact-form matrices, root likelihood vectors from the row-per-thread kernel with a separate multiply and add per
    // term, then on the host, with the host's libm, exp(log L + log prior), the maximum, and the sum of logs in family
    // order (cafe/lambda.cpp:657-724 line by line): the score carries the reference build's bits, so the Nelder-Mead
    // trajectory is the reference's to the last digit.  One GPU, no error model; far slower than the matrix cores
    // (a proof and a debugging aid, not a mode to search in).
    bool opt_objective_reference = false;
    std::vector<int32_t> device_rows;   // the uploaded table in tree-leaf order (kept only for that mode)
    // the pruning kernel switched to the reference's arithmetic for a scope, and back to what the user had chosen (not to the
    // default: CAFEHIP_K2 or a cafehost_set_option "k2" must survive a report)
    struct K2Reference {
        cafehip_ctx* c;
        bool active;
        char before[16];
        K2Reference(cafehip_ctx* ctx, bool on) : c(ctx), active(on)
        {
            if (!active) return;
            if (cafehip_get_option(c, "k2", before, sizeof before) != 0 || cafehip_set_option(c, "k2", "v1ref") != 0)
                throw std::runtime_error(std::string("cafehip: ") + cafehip_last_error());
        }
        ~K2Reference()
        {
            if (active) (void)cafehip_set_option(c, "k2", before);
        }
    };
    struct ReportArith : K2Reference {   // scope guard around the likelihood launches of a report-phase command
        explicit ReportArith(cafehost_session* s_) : K2Reference(s_->ctx, s_->opt_report_reference) {}
    };

    bool speculation_pays()
    {
        if (exchange || (native_comm && !solo)) return false;   // sharded: every evaluation already ends in an exchange
        if (opt_objective_reference) return false;
        if (opt_speculate >= 0) return opt_speculate != 0;
        int wg = 0, cu = 0;
        if (cafehip_launch_info(ctx, &wg, &cu) != 0) return false;
        return wg > 0 && 2 * wg <= cu;
    }

    // (sharded job) would the WHOLE table be more than one round of walk workgroups on one GPU?  Then cutting it N ways pays.
    bool whole_table_fills_a_gpu()
    {
        int wg = 0, cu = 0;
        if (cafehip_launch_info(ctx, &wg, &cu) != 0 || wg <= 0) return true;
        return (long long)wg * std::max(shard_world, 1) > (long long)cu;
    }

    void prefetch_points(const std::vector<std::vector<double>>& pts)
    {
        spec.clear();
        if (pts.size() < 2 || !speculation_pays()) return;
        std::vector<const std::vector<double>*> use;
        const int ncheck = has_mu ? num_params : num_lambdas;
        for (auto& x : pts) {
            bool ok = true;
            for (int i = 0; i < ncheck; ++i) ok = ok && !(x[i] < 0);   // objective() never evaluates these
            for (auto* y : use) ok = ok && (*y != x);
            if (ok && (int)use.size() < CAFEHIP_MAX_SETS) use.push_back(&x);
        }
        if (use.size() < 2) return;
        const int n_sets = (int)use.size();
        std::vector<double> nl((size_t)n_sets * tree.n), nm((size_t)n_sets * tree.n), one_l, one_m;
        for (int q = 0; q < n_sets; ++q) {
            node_rates(use[q]->data(), one_l, one_m);
            std::copy(one_l.begin(), one_l.end(), nl.begin() + (size_t)q * tree.n);
            std::copy(one_m.begin(), one_m.end(), nm.begin() + (size_t)q * tree.n);
        }
        std::vector<double> scores(n_sets);
        std::vector<int32_t> zeros(n_sets);
        if (cafehip_eval_posterior_multi(ctx, n_sets, nl.data(), nm.data(), prior.data(), scores.data(), zeros.data()) != 0)
            return;   // e.g. too many distinct matrices for one pass: the points are evaluated on demand
        ++spec_launches;
        spec_points += n_sets;
        for (int q = 0; q < n_sets; ++q) spec.push_back(SpecEntry{*use[q], scores[q], zeros[q] >= 0 ? zeros[q] + shard_lo : -1});
    }

    // ---- matrices ahead of time (round 5) ------------------------------------------------------------------------------
    // Where whole evaluations are not batched (a table that fills the chip), the points the optimiser may ask for AFTER the
    // evaluation it is about to start (FMinSearch::lookahead) are handed to the library, which builds their transition
    // matrices on a second stream beside that evaluation's pruning (cafehip_prefetch_matrices): the evaluation that then
    // asks for one of them starts at the pruning.  Values, call order and log lines are those of the plain loop.
    int opt_lookahead = -1;      // cafehost_set_option "lookahead": -1 auto, 0 off, 1 on
    long look_calls = 0, look_points = 0;
    bool lookahead_pays()
    {
        if (exchange || opt_objective_reference) return false;   // (the callback path evaluates asynchronously: it builds on demand)
        if (opt_lookahead >= 0) return opt_lookahead != 0;
        // Worth it where an evaluation is a chain of launches, not work: the walk is at most two rounds of workgroups.  There
        // the matrix build is 10-15 % of the chain (configs[1] search -2.4 %, the reference's test1 table -7.9 %); on a table
        // that fills the chip it is 2-3 %, two or three candidate sets are built per evaluation and the builds behind the
        // score kernel outlast the host's turn-around: configs[2] / [3] searches measured 3.3 % SLOWER
        // (profiles/r05_lookahead_ab.txt).
        int wg = 0, cu = 0;
        if (cafehip_launch_info(ctx, &wg, &cu) != 0) return false;
        return wg > 0 && wg <= 2 * cu;
    }
    void lookahead_points(const std::vector<std::vector<double>>& pts)
    {
        if (pts.empty() || !lookahead_pays() || speculation_pays()) return;
        std::vector<const std::vector<double>*> use;
        const int ncheck = has_mu ? num_params : num_lambdas;
        for (auto& x : pts) {
            bool ok = true;
            for (int i = 0; i < ncheck; ++i) ok = ok && !(x[i] < 0);   // objective() never evaluates these
            for (auto* y : use) ok = ok && (*y != x);
            if (ok && (int)use.size() < CAFEHIP_MAX_SETS) use.push_back(&x);
        }
        if (use.empty()) return;
        const int n_sets = (int)use.size();
        look_l.resize((size_t)n_sets * tree.n);
        look_m.resize((size_t)n_sets * tree.n);
        for (int q = 0; q < n_sets; ++q) {
            node_rates(use[q]->data(), look_one_l, look_one_m);
            std::copy(look_one_l.begin(), look_one_l.end(), look_l.begin() + (size_t)q * tree.n);
            std::copy(look_one_m.begin(), look_one_m.end(), look_m.begin() + (size_t)q * tree.n);
        }
        // parked until the next evaluation's own launches are out
        if (cafehip_prefetch_matrices(ctx, n_sets, look_l.data(), look_m.data(), CAFEHIP_PREFETCH_BEHIND_NEXT_EVALUATION) != 0) return;
        ++look_calls;
        look_points += n_sets;
    }
    std::vector<double> look_l, look_m, look_one_l, look_one_m;

    // reset_birthdeath_cache + get_posterior over the WHOLE table (cafe/cafe_main.c:319-326, cafe/lambda.cpp:691-724):
    // on one GPU one synchronous call; sharded, this rank's partial sums stay on the device and the exchange (native
    // RCCL or the caller's callback) produces the global score.  zero = global index of the first zero-likelihood
    // family or -1.
    double evaluate(const std::vector<double>& nl, const std::vector<double>& nm, const std::vector<double>& pr, int32_t& zero)
    {
        double score = 0;
        zero = -1;
        if (native_comm && !solo) {
            // K1 -> tables -> walk -> score kernel with the exchange inside (or one ncclAllGather behind it): the map +
            // sum of cafe/lambda.cpp:698-722 over every rank's block, same bits on every rank
            hip_check(cafehip_eval_posterior_sharded(ctx, nl.data(), nm.data(), pr.data(), &score, &zero));
        } else if (exchange) {
            hip_check(cafehip_eval_posterior_async(ctx, nl.data(), nm.data(), pr.data(), d_exch_chunks, d_exch_fz));
            int z = -1;
            score = exchange(exchange_user, &z);
            zero = z;
        } else if (opt_objective_reference) {
            score = evaluate_reference(nl, nm, pr, zero);
        } else {
            hip_check(cafehip_eval_posterior(ctx, nl.data(), nm.data(), pr.data(), &score, &zero, nullptr, nullptr, nullptr));
            if (zero >= 0) zero += shard_lo;
        }
        return score;
    }

    // get_posterior in the reference's arithmetic and order (option objective_arith=reference, see above)
    double evaluate_reference(const std::vector<double>& nl, const std::vector<double>& nm, const std::vector<double>& pr, int32_t& zero)
    {
        if (err_mfs >= 0) throw std::runtime_error("objective_arith=reference does not cover an error model (the batch entry point scores plain leaves)");
        if (shard_world > 1) throw std::runtime_error("objective_arith=reference runs on one GPU");
        const int F = fam.F(), nleaves = tree.n_leaves(), R = range.root_max - range.root_min + 1;
        if ((int)device_rows.size() != F * nleaves) {
            device_families_current = false;   // (the option was switched on after the upload)
            upload();
        }
        K2Reference restore(ctx, true);
        reset_cache_exact(nl, nm);
        double score = 0;
        zero = -1;
        const int block = 16384;
        std::vector<int32_t> lo(block, range.root_min), hi(block, range.root_max), cm(block, range.max);
        std::vector<double> like((size_t)block * R), posterior(R);
        for (int f0 = 0; f0 < F; f0 += block) {
            const int nb = std::min(block, F - f0);
            hip_check(cafehip_eval_root_likelihoods(ctx, nb, device_rows.data() + (size_t)f0 * nleaves, lo.data(), hi.data(), cm.data(), like.data()));
            for (int i = 0; i < nb; ++i) {
                const double* L = like.data() + (size_t)i * R;
                double max_lik = L[0];                      // __max, libcommon/mathfunc.c
                for (int j = 1; j < R; ++j)
                    if (L[j] > max_lik) max_lik = L[j];
                for (int j = 0; j < R; ++j) posterior[j] = std::exp(std::log(L[j]) + std::log(pr[j]));   // cafe/lambda.cpp:681
                const double max_post = *std::max_element(posterior.begin(), posterior.end());
                if (max_lik == 0) {                         // :715-720 (the reference throws at the first such family)
                    zero = f0 + i;
                    return score;
                }
                score += std::log(max_post);                // :721
            }
        }
        return score;
    }

    // __cafe_best_lambda_search (cafe/lambda.cpp:726-769) / cafe_best_lambda_mu_search (cafe/lambdamu.cpp:323-367)
    double objective(const double* x)
    {
        double score = 0;
        bool skip = false;
        const int ncheck = has_mu ? num_params : num_lambdas;
        for (int i = 0; i < ncheck; ++i)
            if (x[i] < 0) {
                skip = true;
                score = std::log(0.0);
                break;
            }
        if (!skip) {
            std::vector<double> nl, nm;
            node_rates(x, nl, nm);
            int32_t zero = -1;
            bool cached = false;
            for (auto& e : spec)
                if ((int)e.x.size() == num_params && std::equal(e.x.begin(), e.x.end(), x)) {
                    score = e.score;
                    zero = e.zero;
                    cached = true;
                    ++spec_hits;
                    break;
                }
            if (!cached) score = evaluate(nl, nm, prior, zero);
            if (zero >= 0) {  // cafe/lambda.cpp:715-720, 753-760
                if (!quiet)
                    fprintf(stderr, "WARNING: Calculated posterior probability for family %s = 0\n", fam.ids[zero].c_str());
                score = std::log(0.0);
            }
        }
        ++n_evals;
        for (int i = 0; i < num_params; ++i) trace.push_back(x[i]);
        trace.push_back(score);
        if (!has_mu) {
            log("Lambda : %s & Score: %f\n", join_double(x, num_lambdas).c_str(), score);
        } else {
            log("Lambda : %s ", join_double(x, num_lambdas).c_str());
            log("Mu : %s & Score: %f\n", join_double(x + num_lambdas, num_mus - (eqbg ? 1 : 0)).c_str(), score);
        }
        log(".");
        return -score;
    }

    // cafe_best_lambda_by_fminsearch (cafe/lambda.cpp:525-647) / best_lambda_mu_by_fminsearch (cafe/lambdamu.cpp:370-474)
    void search()
    {
        const int max_runs = 10;
        std::vector<double> scores;
        bool converged = false;
        int runs = 0;
        const auto t0 = std::chrono::steady_clock::now();
        do {
            // input_values_randomize, cafe/cafe_main.c:124-163 (k == 0 branch)
            const double mbl = tree.max_branch_length();
            params.assign(num_params, 0.0);
            for (int i = 0; i < num_lambdas; ++i) params[i] = 1.0 / mbl * unifrnd();
            const int mu_len = has_mu ? (num_mus - (eqbg ? 1 : 0)) : 0;
            // the reference draws param->num_mus values here even with -eqbg (one more than it keeps)
            for (int i = 0; i < (has_mu ? num_mus : 0); ++i) {
                const double r = 1.0 / mbl * unifrnd();
                if (i < mu_len) params[num_lambdas + i] = r;
            }
            FMinSearch pfm;
            pfm.init(num_params);
            pfm.tolx = 1e-6;
            pfm.tolf = 1e-6;
            pfm.eq = [&](const double* x) { return objective(x); };
            pfm.prefetch = [&](const std::vector<std::vector<double>>& pts) { prefetch_points(pts); };
            pfm.lookahead = [&](const std::vector<std::vector<double>>& pts) { lookahead_points(pts); };
            std::vector<double> start = params;
            pfm.minimize(start.data());
            params = pfm.v[0];
            spec.clear();
            search_iters = pfm.iters;
            last_score = pfm.fv[0];
            log("\n");
            log("Lambda Search Result: %d\n", pfm.iters);
            if (!has_mu) {
                log("Lambda : %s & Score: %f\n", join_double(params.data(), num_lambdas).c_str(), pfm.fv[0]);
            } else {
                log("Lambda : %s & Score: %f", join_double(params.data(), num_lambdas).c_str(), pfm.fv[0]);
                log("Mu : %s & Score: %f\n", join_double(params.data() + num_lambdas, mu_len).c_str(), pfm.fv[0]);
            }
            if (runs > 0) {
                const double minscore = *std::min_element(scores.begin(), scores.end());
                if (std::fabs(minscore - pfm.fv[0]) < 10 * pfm.tolf) converged = true;
            }
            scores.push_back(pfm.fv[0]);
            ++runs;
        } while (checkconv && !converged && runs < max_runs);
        search_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (checkconv) {
            if (converged)
                log("score converged in %d runs.\n", runs);
            else
                log("score failed to converge in %d runs.\n", max_runs);
        }
    }

    // __cafe_cmd_lambda_tree, cafe/cafe_shell.c:334-393
    void set_lambda_tree(const std::string& text)
    {
        HostTree lt = HostTree::parse(text);
        if (lt.n != tree.n) throw std::runtime_error("Lambda has a different topology from the tree");
        node_class.assign(lt.n, -1);
        int named = 0;
        std::vector<int> seen;
        for (int i = 0; i < lt.n; ++i) {
            int taxaid = 0;
            if (!lt.name[i].empty()) {
                sscanf(lt.name[i].c_str(), "%d", &taxaid);
                ++named;
            }
            node_class[i] = taxaid - 1;  // phylogeny_lambda_parse_func :324-332
            if (node_class[i] >= 0 && std::find(seen.begin(), seen.end(), node_class[i]) == seen.end())
                seen.push_back(node_class[i]);
        }
        if (named != lt.n - 1)
            throw std::runtime_error("ERROR(lambda -t): Branch lambda classes not totally specified.\n" + text +
                                     "\nYou have to specify lambda classes for all branches including the internal "
                                     "branches of the tree.");
        for (int c : seen)
            if (c >= (int)seen.size()) throw std::runtime_error("lambda classes must be numbered 1.." + std::to_string(seen.size()));
        lambda_tree = lt;
        have_lambda_tree = true;
        num_lambdas = (int)seen.size();
        if (!quiet && shard_rank == 0) {   // (every rank runs the script; one echo)
            printf("The number of lambdas is %d\n", num_lambdas);
            fflush(stdout);
        }
        log("Lambda Tree: %s\n", text.c_str());
    }

    void prereqs(bool need_family, bool need_tree)
    {
        if (need_family && !have_family) throw std::runtime_error("ERROR: The gene families were not loaded. Please load gene families with the 'load' command.\n");
        if (need_tree && !have_tree) throw std::runtime_error("ERROR: The tree was not loaded. Please load a tree with the 'tree' command.\n");
    }

    // ---- commands ----------------------------------------------------------------------
    // cafe_family_filter, cafe/gene_family.cpp:273-353: keep a family only when BOTH subtrees below the
    // root hold at least one gene (parsimony: at least one copy at the root); ranges follow the new maximum
    void family_filter()
    {
        const int ns = (int)fam.species.size();
        const int F = fam.F();
        std::vector<char> mark(tree.n);
        std::vector<std::string> ids, desc;
        std::vector<int32_t> counts;
        int mx = 0;
        for (int i = 0; i < F; ++i) {
            std::fill(mark.begin(), mark.end(), 0);
            for (int s_ = 0; s_ < ns; ++s_) {
                if (species_index[s_] < 0 || fam.counts[(size_t)i * ns + s_] <= 0) continue;
                for (int p = species_index[s_]; p >= 0 && !mark[p]; p = tree.parent[p]) mark[p] = 1;
            }
            if (mark[tree.left[tree.root]] && mark[tree.right[tree.root]]) {
                ids.push_back(fam.ids[i]);
                desc.push_back(fam.desc[i]);
                counts.insert(counts.end(), fam.counts.begin() + (size_t)i * ns, fam.counts.begin() + (size_t)(i + 1) * ns);
                for (int s_ = 0; s_ < ns; ++s_) mx = std::max(mx, (int)fam.counts[(size_t)i * ns + s_]);
            }
        }
        if ((int)ids.size() != F) {
            log("The Number of families : %d ==> %d\n", F, (int)ids.size());
            fam.ids.swap(ids);
            fam.desc.swap(desc);
            fam.counts.swap(counts);
        }
        if (fam.max_size != mx) {
            fam.max_size = mx;
            range = init_family_size(mx);
        }
        device_families_current = false;
    }

    int cmd_load(const std::vector<std::string>& tokens)
    {  // cafe_cmd_load, cafe/cafe_commands.cpp:868-970; args :817-866
        auto args = build_argument_list(tokens);
        std::string file;
        int max_size = -1;
        bool filter = false;
        for (auto& a : args) {
            if (a.opt == "-t" && !a.argv.empty()) num_threads = atoi(a.argv[0].c_str());
            if (a.opt == "-r" && !a.argv.empty()) num_random_samples = atoi(a.argv[0].c_str());
            if (a.opt == "-max_size" && !a.argv.empty()) max_size = atoi(a.argv[0].c_str());
            if (a.opt == "-p" && !a.argv.empty()) pvalue = atof(a.argv[0].c_str());
            if (a.opt == "-l" && !a.argv.empty() && a.argv[0] != "stdout") {
                std::string name;
                for (size_t i = 0; i < a.argv.size(); ++i) name += (i ? " " : "") + a.argv[i];
                FILE* f = fopen(name.c_str(), "a");
                if (!f) throw std::runtime_error("ERROR(load): Cannot open log file: " + name);
                if (own_log) fclose(flog);
                flog = f;
                own_log = true;
                log_name = name;
            }
            if (a.opt == "-filter") filter = true;
            if (a.opt == "-i") {
                for (size_t i = 0; i < a.argv.size(); ++i) file += (i ? " " : "") + a.argv[i];
            }
        }
        if (file.empty()) throw std::runtime_error("Usage(load): load <family file>");
        fam.load(file, max_size);
        have_family = true;
        cond_dist.clear();
        rep_sizes.clear();   // (`report <name> save` has nothing to write for a new table)
        rep_max_p.clear();
        rep_expand.clear();
        err_file.clear();
        err_mfs = -1;
        range = init_family_size(fam.max_size);  // set_range_from_family
        sync_species_index();
        if (filter && !have_tree) {
            fprintf(stderr, "Error(load): You did not specify tree. Skip filtering\n");
        } else if (filter) {
            family_filter();
        }
        device_families_current = false;
        // log_param_values, cafe/cafe_commands.cpp:1918-1943
        log("-----------------------------------------------------------\n");
        log("Family information: %s\n", file.c_str());
        log("Log: %s\n", log_name.c_str());
        if (have_tree) log("Tree: %s\n", tree.newick.c_str());
        log("The number of families is %d\n", fam.F());
        log("Root Family size : %d ~ %d\n", range.root_min, range.root_max);
        log("Family size : %d ~ %d\n", range.min, range.max);
        log("P-value: %g\n", pvalue);
        log("Num of Threads: %d\n", num_threads);
        log("Num of Random: %d\n", num_random_samples);
        return 0;
    }

    int cmd_tree(const std::vector<std::string>& tokens)
    {  // cafe_cmd_tree, cafe/cafe_commands.cpp:1127-1190
        std::string newick;
        if (tokens.size() > 2 && tokens[1] == "-i") {
            std::ifstream in(tokens[2]);
            if (!in) throw std::runtime_error("Failed to read file '" + tokens[2] + "'");
            std::stringstream ss;
            ss << in.rdbuf();
            newick = ss.str();
        } else if (tokens.size() >= 2) {
            for (size_t i = 1; i < tokens.size(); ++i) newick += tokens[i];
        } else {
            throw std::runtime_error("Usage(tree): tree <newick>");
        }
        tree = HostTree::parse(newick);
        have_tree = true;
        have_lambda_tree = false;
        rep_sizes.clear();
        rep_expand.clear();
        if (!quiet) log("%s\n", tree.newick.c_str());
        sync_species_index();
        device_families_current = false;
        return 0;
    }

    static std::vector<double> doubles_of(const Argument& a)
    {
        std::vector<double> out;
        for (auto& s : a.argv) out.push_back(atof(s.c_str()));
        return out;
    }

    // ---- k-cluster model ---------------------------------------------------------------------------------
    // Rates of cluster k on every node: initialize_k_bd -> set_birth_death_probabilities4 (cafe/cafe_shell.c:148-177,
    // 194-213): the K (or, with -f, K - 1) values of the node's lambda class sit next to each other in the
    // parameter vector; with -f cluster 0 has lambda 0 (an identity matrix: nothing changes along the tree).
    void cluster_node_rates(const double* x, int k, double* nl, double* nm) const
    {
        const int K = k_clusters, fix = fixcluster0 ? 1 : 0;
        for (int i = 0; i < tree.n; ++i) {
            int cls = have_lambda_tree ? node_class[i] : 0;
            if (cls < 0) cls = 0;
            nl[i] = fix ? (k == 0 ? 0.0 : x[cls * (K - 1) + (k - 1)]) : x[cls * K + k];
            nm[i] = -1;
        }
    }

    // __cafe_cluster_lambda_search (cafe/cafe_main.c:272-308): weights from the parameters
    // (input_values_copy_weights, libtree/input_values.c:84-94), cafe_get_clustered_posterior on the device in ONE
    // pass over the table for all K clusters, then the weights become the mean memberships (cafe_main.c:243-245).
    double cluster_objective(const double* x)
    {
        // the device call scores the table that is on THIS device: a shard would give every rank a different score
        // and memberships indexed by local family
        if (exchange || native_comm || shard_world > 1)
            throw std::runtime_error("the k-cluster model is not sharded: run `lambda -k` / `score` of a clustered model on one rank");
        const int K = k_clusters, fix = fixcluster0 ? 1 : 0;
        const int n_lam = num_lambdas * (K - fix);
        double score = 0;
        bool skip = false;
        for (int i = 0; i < num_params; ++i)
            if (x[i] < 0) {
                skip = true;
                score = std::log(0.0);
                break;
            }
        if (!skip) {
            k_weights.assign(K, 0.0);
            double sumofweights = 0;
            for (int i = 0; i < K - 1; ++i) {
                k_weights[i] = x[n_lam + i];
                sumofweights += x[n_lam + i];
            }
            k_weights[K - 1] = 1 - sumofweights;
            std::vector<double> nl((size_t)K * tree.n), nm((size_t)K * tree.n);
            for (int k = 0; k < K; ++k) cluster_node_rates(x, k, nl.data() + (size_t)k * tree.n, nm.data() + (size_t)k * tree.n);
            int32_t zero = -1;
            last_membership_sums.assign(K, 0.0);
            p_z_membership.assign((size_t)std::max(fam.F(), 1) * K, 0.0);
            hip_check(cafehip_eval_clustered_posterior(ctx, K, nl.data(), nm.data(), k_weights.data(), prior.data(), &score, &zero,
                                                       last_membership_sums.data(), nullptr, p_z_membership.data()));
            if (zero >= 0 && !quiet)
                fprintf(stderr, "WARNING: Calculated posterior probability for family %s = 0\n", fam.ids[zero].c_str());
            for (int k = 0; k < K; ++k) k_weights[k] = last_membership_sums[k] / fam.F();
        }
        ++n_evals;
        for (int i = 0; i < num_params; ++i) trace.push_back(x[i]);
        trace.push_back(score);
        if (!quiet && shard_rank == 0) {
            printf("Lambda : %s\n", join_double(x, n_lam).c_str());
            printf("p : %s\n", join_double(k_weights.data(), K).c_str());
            printf("Score: %f\n", score);
            printf(".");
            fflush(stdout);
        }
        return -score;
    }

    // cafe_best_lambda_by_fminsearch with k > 0 (cafe/lambda.cpp:525-647)
    void cluster_search()
    {
        if (exchange || native_comm) throw std::runtime_error("the k-cluster search is not sharded: run it on one rank");
        const int K = k_clusters, fix = fixcluster0 ? 1 : 0, kfix = K - fix;
        const int n_lam = num_lambdas * kfix;
        const int max_runs = 10;
        std::vector<double> scores;
        bool converged = false;
        int runs = 0;
        const auto t0 = std::chrono::steady_clock::now();
        do {
            // input_values_randomize, k > 0 branch (cafe/cafe_main.c:129-151)
            const double mbl = tree.max_branch_length();
            params.assign(num_params, 0.0);
            for (int i = 0; i < n_lam; ++i) params[i] = 1.0 / mbl * unifrnd();
            k_weights.assign(K, 0.0);
            double sumw = 0;
            for (int j = 0; j < K; ++j) {
                k_weights[j] = unifrnd();
                sumw += k_weights[j];
            }
            for (int j = 0; j < K; ++j) k_weights[j] = k_weights[j] / sumw;
            for (int j = 0; j < K - 1; ++j) params[n_lam + j] = k_weights[j];
            FMinSearch pfm;
            pfm.init(num_params);
            pfm.tolx = 1e-5;
            pfm.tolf = 1e-5;
            pfm.eq = [&](const double* x) { return cluster_objective(x); };
            std::vector<double> start = params;
            pfm.minimize(start.data());
            params = pfm.v[0];
            double current_p = params[n_lam], prev_p;
            do {   // restart from the mean memberships of the last evaluation until the first weight settles (:571-591)
                for (int j = 0; j < K - 1; ++j) params[n_lam + j] = last_membership_sums[j] / fam.F();
                std::vector<double> again = params;
                pfm.minimize(again.data());
                params = pfm.v[0];
                prev_p = current_p;
                current_p = params[n_lam];
            } while (current_p - prev_p > pfm.tolx);
            search_iters = pfm.iters;
            last_score = pfm.fv[0];
            log("\n");
            log("Lambda Search Result: %d\n", pfm.iters);
            log("Lambda : %s%s\n", fix ? "0," : "", join_double(params.data(), n_lam).c_str());
            log("p : %s\n", join_double(k_weights.data(), K).c_str());
            log("p0 : %f\n", params[n_lam]);
            log("Score: %f\n", pfm.fv[0]);
            if (runs > 0) {
                const double minscore = *std::min_element(scores.begin(), scores.end());
                if (std::fabs(minscore - pfm.fv[0]) < 10 * pfm.tolf) converged = true;
            }
            scores.push_back(pfm.fv[0]);
            ++runs;
        } while (checkconv && !converged && runs < max_runs);
        search_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (checkconv) {
            if (converged)
                log("score converged in %d runs.\n", runs);
            else
                log("score failed to converge in %d runs.\n", max_runs);
        }
    }

    // log_cluster_membership, cafe/gene_family.cpp:355-372
    void log_cluster_membership()
    {
        std::ostringstream os;
        os << "The Number of families : " << fam.F() << "\n";
        for (int i = 0; i < fam.F(); ++i) {
            os << "family " << fam.ids[i] << ": ";
            for (int k = 0; k < k_clusters; ++k) os << " " << p_z_membership[(size_t)i * k_clusters + k];
            os << "\n";
        }
        log("%s", os.str().c_str());
    }

    void finish_command(const std::vector<std::string>& tokens, const char* what)
    {
        if (k_clusters > 0) {
            log("DONE: %s Search or setting, for command:\n", what);
            std::string cmd;
            for (auto& t : tokens) cmd += t + " ";
            log("%s\n", cmd.c_str());
            return;
        }
        // after lambda/lambdamu the cache is rebuilt for the final parameters and left resident
        // (cafe/lambda.cpp:504-507, cafe/lambdamu.cpp:254-257)
        std::vector<double> nl, nm;
        node_rates(params.data(), nl, nm);
        bool ok = true;
        for (double v : nl) ok = ok && v >= 0;
        if (ok) hip_check(cafehip_reset_birthdeath_cache(ctx, nl.data(), nm.data()));
        log("DONE: %s Search or setting, for command:\n", what);
        std::string cmd;
        for (auto& t : tokens) cmd += t + " ";
        log("%s\n", cmd.c_str());
    }

    // ---- lambda -s -e: one Nelder-Mead search per family (cafe_each_best_lambda_by_fminsearch,
    //      cafe/lambda.cpp:911-1010; objective __cafe_each_best_lambda_search :872-908) ----
    // Every family is scored under ITS OWN ranges (cafe_family_set_size_with_family_forced,
    // cafe/cafe_family.c:236-255), so the matrices are rebuilt per family at that size: a one-family table on
    // a second device context (the "tree-level" seam of SURVEY.md section 8b routed to the batch API with F = 1).
    // Objective = log(max_i L_root[i]) -- no prior.  Quirks kept: the optimiser object is reused across
    // families, and because cafe_shell_set_lambda aliases param->lambda to the last parameters set
    // (cafe/cafe_shell.c:245-251), family i+1 starts from family i's optimum (the first from 0.5 / max branch).
    std::vector<std::vector<double>> each_lambda;  // per family fitted parameters
    void each_family_search(const std::string& outfile)
    {
        if (shard_world > 1) throw std::runtime_error("lambda -e runs on one rank (it is a loop of one-family searches)");
        if (!ctx_one && cafehip_create(&ctx_one, device_id) != 0)
            throw std::runtime_error(std::string("cafehip: ") + cafehip_last_error());
        hip_check(cafehip_set_tree(ctx_one, tree.n, tree.parent.data(), tree.left.data(), tree.right.data(), tree.bl.data()));
        const int F = fam.F(), nl = tree.n_leaves(), ns = (int)fam.species.size();
        // ref: first identical row (cafe/cafe_family.c:9-34)
        std::vector<int> ref(F, -1);
        {
            std::map<std::vector<int32_t>, int> first;
            for (int i = 0; i < F; ++i) {
                std::vector<int32_t> row(fam.counts.begin() + (size_t)i * ns, fam.counts.begin() + (size_t)(i + 1) * ns);
                auto it = first.find(row);
                if (it == first.end()) first.emplace(std::move(row), i);
                else ref[i] = it->second;
            }
        }
        const double mbl = tree.max_branch_length();
        std::vector<double> start(num_lambdas, 0.5 / mbl);
        FMinSearch pfm;
        pfm.init(num_lambdas);
        pfm.tolx = 1e-6;
        pfm.tolf = 1e-6;
        std::vector<double> one_prior(1000, 1.0);  // unused by the max-likelihood output, must be positive
        pfm.eq = [&](const double* x) {
            double score = 0;
            bool skip = false;
            for (int i = 0; i < num_lambdas; ++i)
                if (x[i] < 0) {
                    skip = true;
                    score = std::log(0.0);
                    break;
                }
            if (!skip) {
                std::vector<double> nlam, nmu;
                node_rates(x, nlam, nmu);
                double sc = 0, ml = 0;
                int32_t zero = -1;
                hip_check(cafehip_eval_posterior(ctx_one, nlam.data(), nmu.data(), one_prior.data(), &sc, &zero, &ml, nullptr, nullptr));
                score = std::log(ml);
            }
            ++n_evals;
            log("\tLambda : %s & Score: %f\n", join_double(x, num_lambdas).c_str(), score);
            log("\n");
            return -score;
        };
        auto family_tree_string = [&](int i, const std::vector<double>& x) {
            std::vector<double> nlam, nmu;
            node_rates(x.data(), nlam, nmu);
            std::vector<int> size(tree.n, -1);
            for (int s_ = 0; s_ < ns; ++s_)
                if (species_index[s_] >= 0) size[species_index[s_]] = fam.counts[(size_t)i * ns + s_];
            return tree_string(
                [&](int v) {  // cafe_tree_string_familysize_lambda, cafe/cafe_tree.c:121-129
                    if (tree.bl[v] <= 0) return std::string();
                    char buf[96];
                    std::string out = tree.name[v];
                    if (size[v] >= 0) {
                        snprintf(buf, sizeof buf, "<%d>", size[v]);
                        out += buf;
                    }
                    snprintf(buf, sizeof buf, "_%lf", nlam[v]);
                    return out + buf;
                },
                true);
        };
        each_lambda.assign(F, {});
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < F; ++i) {
            if (ref[i] >= 0) {
                each_lambda[i] = each_lambda[ref[i]];
                start = each_lambda[i];  // cafe_shell_set_lambdas(param, pitem->lambda) re-aliases param->lambda
                log("%s: Lambda Search Result of %d/%d in %d iteration \n", fam.ids[i].c_str(), i + 1, F, pfm.iters);
                log("%s: %s\n", fam.ids[i].c_str(), family_tree_string(i, each_lambda[i]).c_str());
                continue;
            }
            std::vector<int32_t> counts(nl, 0);
            int mx = 0;
            for (int s_ = 0; s_ < ns; ++s_) {
                if (species_index[s_] < 0) continue;
                const int c = fam.counts[(size_t)i * ns + s_];
                counts[species_index[s_] / 2] = c;
                mx = std::max(mx, c);
            }
            const int root_max = (int)std::rint(mx * 1.25), range_max = mx + std::max(50, mx / 5);
            log("%s:\n", fam.ids[i].c_str());
            if (root_max < 1) {
                // all counts zero: root range [1, 0] is empty.  The reference then maximises over an empty root
                // vector and reads the stale element 0 left by the previous evaluation (libcommon/mathfunc.c:26-38
                // with size 0), so its "search" fits leftover memory.  Not reproducible and not meaningful:
                // the family keeps the lambda the search would have started from.
                each_lambda[i] = start;
                log("Lambda Search Result of %d/%d in 0 iteration (empty root range: not searched)\n", i + 1, F);
                log("%s\n", family_tree_string(i, each_lambda[i]).c_str());
                continue;
            }
            hip_check(cafehip_set_families(ctx_one, 1, nl, counts.data(), nullptr, 0, range_max, 1, root_max));
            if (err_mfs >= 0)
                hip_check(cafehip_set_error_model(ctx_one, err_mfs, err_matrix.data(), err_leaf.data()));
            pfm.minimize(start.data());
            each_lambda[i] = pfm.v[0];
            start = each_lambda[i];
            bool near_boundary = false;
            for (int j = 0; j < num_lambdas; ++j) {
                const double a = each_lambda[i][j] * mbl;
                if (a >= 0.5 || std::fabs(a - 0.5) < 1e-3) near_boundary = true;
            }
            log("Lambda Search Result of %d/%d in %d iteration \n", i + 1, F, pfm.iters);
            if (near_boundary) log("Caution : at least one lambda near boundary\n");
            if (near_boundary) log("@@ ");
            log("%s\n", family_tree_string(i, each_lambda[i]).c_str());
        }
        search_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        search_iters = pfm.iters;
        params = F ? each_lambda[F - 1] : std::vector<double>(num_lambdas, 0.5 / mbl);
        last_score = F ? pfm.fv[0] : 0.0;
        // per-family result table: cafe/lambda.cpp:465-497 (`-e` implies write_files; <outfile>.lambda, the
        // .html twin is presentation and not written)
        FILE* fp = stdout;
        if (!outfile.empty()) {
            fp = fopen((outfile + ".lambda").c_str(), "w");
            if (!fp) throw std::runtime_error("Cannot open file: " + outfile + ".lambda");
        }
        for (int i = 0; i < F; ++i) {
            for (int j = 0; j < num_lambdas; ++j) {
                const double a = each_lambda[i][j] * mbl;
                if (a >= 0.5 || std::fabs(a - 0.5) < 1e-3) {
                    fprintf(fp, "@@ ");
                    break;
                }
            }
            fprintf(fp, "%s\t%s\n", fam.ids[i].c_str(), family_tree_string(i, each_lambda[i]).c_str());
        }
        if (fp != stdout) fclose(fp);
        else fflush(stdout);
    }

    int cmd_lambda(const std::vector<std::string>& tokens)
    {  // cafe_cmd_lambda, cafe/lambda.cpp:369-515
        prereqs(true, true);
        auto args = build_argument_list(tokens);
        has_mu = false;
        eqbg = false;
        num_mus = 0;
        checkconv = false;
        have_lambda_tree = false;
        num_lambdas = 1;
        bool search_flag = false, score_flag = false, each = false;
        double vlambda = -1;
        std::vector<double> lambdas, weights_arg;
        k_clusters = 0;
        fixcluster0 = false;
        struct LambdaRange { double start, step, end; };
        std::vector<LambdaRange> ranges;
        std::string outfile;
        for (auto& a : args) {
            if (a.opt == "-s") search_flag = true;
            else if (a.opt == "-checkconv") checkconv = true;
            else if (a.opt == "-score") score_flag = true;
            else if (a.opt == "-t") {
                if (a.argv.empty()) throw std::runtime_error("lambda -t needs a lambda tree");
                set_lambda_tree(a.argv.back());
            } else if (a.opt == "-l") lambdas = doubles_of(a);
            else if (a.opt == "-p") weights_arg = doubles_of(a);   // k_weights (cafe/lambda.cpp:129-133)
            else if (a.opt == "-k") {
                if (!a.argv.empty()) k_clusters = atoi(a.argv[0].c_str());
                if (k_clusters < 1 || k_clusters > CAFEHIP_MAX_SETS)
                    throw std::runtime_error("lambda -k: 1.." + std::to_string(CAFEHIP_MAX_SETS) + " clusters");
            } else if (a.opt == "-f") fixcluster0 = true;
            else if (a.opt == "-v") {
                // SINGLE_LAMBDA: set_all_lambdas before the command (cafe/lambda.cpp:386-389); a search starts from
                // random values anyway, so it only matters for the set form
                if (!a.argv.empty()) vlambda = atof(a.argv[0].c_str());
            } else if (a.opt == "-r") {
                for (auto& t : a.argv) {  // start:step:end per lambda, cafe/lambda.cpp:158-166
                    LambdaRange r{0, 0, 0};
                    sscanf(t.c_str(), "%lf:%lf:%lf", &r.start, &r.step, &r.end);
                    ranges.push_back(r);
                }
            } else if (a.opt == "-e") {
                each = true;
            } else if (a.opt == "-o") {
                if (!a.argv.empty()) outfile = a.argv[0];
            } else
                throw std::runtime_error("lambda " + a.opt + " is outside this build's scope (supported: -s -l -v -t -r -o -e -score -checkconv -k -f -p)");
        }
        upload_beside_prior_fit();
        n_evals = 0;
        trace.clear();
        if (!ranges.empty()) {
            // likelihood surface on a grid: cafe/lambda.cpp:391-418, cafe_lambda_distribution :192-231,
            // write_lambda_distribution :233-257 (first range varies slowest, gmatrix_dim_index)
            FILE* fp = nullptr;
            if (!outfile.empty() && !(fp = fopen(outfile.c_str(), "w")))
                throw std::runtime_error("ERROR(lambda): Cannot open file: " + outfile);
            set_prior_rfsize_empirical();
            num_lambdas = (int)ranges.size();
            num_params = num_lambdas;
            for (size_t j = 0; j < ranges.size(); ++j)
                log("%zust Distribution: %s : %s : %s\n", j + 1, fmt_g(ranges[j].start).c_str(), fmt_g(ranges[j].step).c_str(),
                    fmt_g(ranges[j].end).c_str());
            std::vector<int> size(ranges.size());
            long long total = 1;
            for (size_t j = 0; j < ranges.size(); ++j) {
                size[j] = 1 + (int)std::rint((ranges[j].end - ranges[j].start) / ranges[j].step);
                total *= size[j];
            }
            std::vector<int> idx(ranges.size());
            std::vector<double> x(ranges.size());
            auto point = [&](long long e, std::vector<double>& y) {
                y.resize(ranges.size());
                for (int j = (int)ranges.size() - 1; j >= 0; --j) {
                    y[j] = ranges[j].step * (double)(e % size[j]) + ranges[j].start;
                    e /= size[j];
                }
            };
            // Sharded job, small table: the grid points are independent evaluations (cafe_lambda_distribution loops over them,
            // cafe/lambda.cpp:192-231) and a table that does not fill one GPU gains nothing from being cut N ways, so rank r
            // evaluates points r, r + N, ... of the grid on the WHOLE table alone (`solo`, as lhtest deals its files), the values
            // are gathered and the loop below finds every point's score ready: the log lines and the file are the one-rank
            // run's byte for byte (the score's bits do not depend on how the table is cut).  A table that fills the chip stays
            // sharded: every point on all ranks is the same work without a second upload.
            std::vector<double> dealt_score;
            std::vector<int32_t> dealt_zero;
            if (native_comm && shard_world > 1 && opt_grid_deal != 0 && total >= 2LL * shard_world && total < (1LL << 24) &&
                (opt_grid_deal > 0 || !whole_table_fills_a_gpu())) {
                const long long per = (total + shard_world - 1) / shard_world;
                std::vector<double> mine((size_t)per * 2, 0.0), all((size_t)per * 2 * shard_world, 0.0);
                std::string problem;
                solo = true;
                device_families_current = false;
                try {
                    upload();
                    std::vector<double> y, nl, nm;
                    const int ncheck = has_mu ? num_params : num_lambdas;
                    for (long long e = shard_rank, k = 0; e < total; e += shard_world, ++k) {
                        point(e, y);
                        bool neg = false;
                        for (int i = 0; i < ncheck && i < (int)y.size(); ++i) neg = neg || y[i] < 0;
                        int32_t zero = -1;
                        double sc = 0;
                        if (!neg) {   // (objective() never evaluates a negative rate)
                            node_rates(y.data(), nl, nm);
                            sc = evaluate(nl, nm, prior, zero);
                        }
                        mine[(size_t)2 * k] = sc;
                        mine[(size_t)2 * k + 1] = (double)zero;
                    }
                } catch (const std::exception& ex) {
                    problem = ex.what();   // the others are about to wait for this rank's values
                }
                solo = false;
                device_families_current = false;   // the next evaluation loads this rank's block again
                std::vector<char> ok(shard_world, 0);
                const char my_ok = problem.empty() ? 1 : 0;
                if (allgather(allgather_user, &my_ok, 1, ok.data(), 1) != 0) throw std::runtime_error("lambda -r: allgather failed");
                for (int r = 0; r < shard_world; ++r)
                    if (!ok[r]) {
                        if (fp) fclose(fp);
                        throw std::runtime_error("lambda -r: rank " + std::to_string(r) + " failed" + (r == shard_rank ? ": " + problem : std::string()));
                    }
                if (allgather(allgather_user, mine.data(), (long long)(mine.size() * sizeof(double)), all.data(), (long long)(mine.size() * sizeof(double))) != 0)
                    throw std::runtime_error("lambda -r: allgather failed");
                dealt_score.resize((size_t)total);
                dealt_zero.resize((size_t)total);
                for (long long e = 0; e < total; ++e) {
                    const size_t at = (size_t)(e % shard_world) * (size_t)per * 2 + (size_t)(e / shard_world) * 2;
                    dealt_score[(size_t)e] = all[at];
                    dealt_zero[(size_t)e] = (int32_t)all[at + 1];
                }
                grid_points_dealt += total;
                if (opt_timing && shard_rank == 0) fprintf(stderr, "lambda -r: %lld grid points dealt to %d ranks\n", total, shard_world);
            }
            for (long long e = 0; e < total; ++e) {
                long long rem = e;
                for (int j = (int)ranges.size() - 1; j >= 0; --j) {
                    idx[j] = (int)(rem % size[j]);
                    rem /= size[j];
                }
                for (size_t j = 0; j < ranges.size(); ++j) x[j] = ranges[j].step * idx[j] + ranges[j].start;
                if (!dealt_score.empty()) {
                    spec.clear();
                    spec.push_back(SpecEntry{x, dealt_score[(size_t)e], dealt_zero[(size_t)e]});
                } else if (e % CAFEHIP_MAX_SETS == 0) {
                    // the next grid points in one batched pass (small tables): same values, one launch
                    std::vector<std::vector<double>> pts;
                    for (long long e2 = e; e2 < total && e2 < e + CAFEHIP_MAX_SETS; ++e2) {
                        long long rem2 = e2;
                        std::vector<double> y(ranges.size());
                        for (int j = (int)ranges.size() - 1; j >= 0; --j) {
                            y[j] = ranges[j].step * (double)(rem2 % size[j]) + ranges[j].start;
                            rem2 /= size[j];
                        }
                        pts.push_back(y);
                    }
                    prefetch_points(pts);
                }
                const double v = -objective(x.data());
                if (fp) {
                    fprintf(fp, "%lf", idx[0] * ranges[0].step + ranges[0].start);
                    for (size_t k = 1; k < ranges.size(); ++k) fprintf(fp, "\t%lf", idx[k] * ranges[k].step + ranges[k].start);
                    fprintf(fp, "\t%lf\n", v);
                }
            }
            if (fp) fclose(fp);
            spec.clear();   // the last batch of grid points must not outlive the command
            return 0;
        }
        set_prior_rfsize_empirical();
        num_params = num_lambdas;
        if (k_clusters == 0 && !weights_arg.empty()) k_clusters = (int)weights_arg.size();   // -p alone sets k (:129-139)
        if (k_clusters > 0) {
            // clustered model: lambda_search / set_parameters, cafe/lambda.cpp:273-352
            const int fix = fixcluster0 ? 1 : 0;
            if (k_clusters - fix < 1) throw std::runtime_error("lambda -k -f needs at least two clusters");
            num_params = num_lambdas * (k_clusters - fix) + (k_clusters - 1);
            if (search_flag) {
                cluster_search();
            } else {
                if ((int)lambdas.size() != num_lambdas * (k_clusters - fix) || (int)weights_arg.size() != k_clusters)
                    throw std::runtime_error("ERROR(lambda): Number of parameters not correct: -l needs " +
                                             std::to_string(num_lambdas * (k_clusters - fix)) + " values and -p " +
                                             std::to_string(k_clusters));
                params = lambdas;   // input_values_set_lambdas + input_values_set_k_weights (the first K - 1 weights)
                params.insert(params.end(), weights_arg.begin(), weights_arg.end() - 1);
                last_score = cluster_objective(params.data());
            }
            if (search_flag) log_cluster_membership();
            finish_command(tokens, "Lambda");
            return 0;
        }
        if (search_flag && each) {
            each_family_search(outfile);
        } else if (search_flag) {
            search();
        } else {
            if (lambdas.empty() && vlambda > 0) lambdas.assign(num_lambdas, vlambda);
            if ((int)lambdas.size() != num_lambdas)
                throw std::runtime_error("ERROR(lambda): Number of parameters not correct. The number of -l lambdas are " +
                                         std::to_string(lambdas.size()) + " they need to be " + std::to_string(num_lambdas));
            params = lambdas;
            if (score_flag) last_score = objective(params.data());
        }
        finish_command(tokens, "Lambda");
        return 0;
    }

    int cmd_lambdamu(const std::vector<std::string>& tokens)
    {  // cafe_cmd_lambdamu, cafe/lambdamu.cpp:218-269
        prereqs(true, true);
        auto args = build_argument_list(tokens);
        has_mu = true;
        eqbg = false;
        checkconv = false;
        have_lambda_tree = false;
        num_lambdas = 1;
        k_clusters = 0;        // a lambda/mu model replaces a clustered one (cafe/lambdamu.cpp:218-269 has no -k)
        fixcluster0 = false;
        bool search_flag = false;
        std::vector<double> lambdas, mus;
        for (auto& a : args) {
            if (a.opt == "-s") search_flag = true;
            else if (a.opt == "-checkconv") checkconv = true;
            else if (a.opt == "-eqbg") eqbg = true;
            else if (a.opt == "-t") {
                if (a.argv.empty()) throw std::runtime_error("lambdamu -t needs a lambda tree");
                set_lambda_tree(a.argv.back());
            } else if (a.opt == "-l") lambdas = doubles_of(a);
            else if (a.opt == "-m") mus = doubles_of(a);
            else
                throw std::runtime_error("lambdamu " + a.opt + " is outside this build's scope (supported: -s -l -m -t -eqbg -checkconv)");
        }
        if (eqbg && !have_lambda_tree)
            throw std::runtime_error("ERROR(lambdamu): Cannot use option eqbg without specifying a lambda tree. \n");
        num_mus = num_lambdas;
        num_params = num_lambdas + num_mus - (eqbg ? 1 : 0);
        upload_beside_prior_fit();
        n_evals = 0;
        trace.clear();
        set_prior_rfsize_empirical();
        if (search_flag) {
            search();
        } else {
            if ((int)(lambdas.size() + mus.size()) != num_params)
                throw std::runtime_error("ERROR(lambdamu): Number of parameters not correct.");
            params = lambdas;
            params.insert(params.end(), mus.begin(), mus.end());
            last_score = objective(params.data());
        }
        finish_command(tokens, "Lamda,Mu");
        return 0;
    }


    // ---- Monte-Carlo null: cafe_conditional_distribution, cafe/conditional_distribution.cpp:10-120 ----
    // Reference order for `-t 1`: root sizes ascending, trials sequential, one unifrnd() per non-root
    // node in prefix order (cafe/cafe_tree.c:533-569); samples are drawn on the host from the device-built
    // matrices, the likelihood of every simulated family is ONE batched GPU call.
    void compute_conditional_distribution(const std::vector<std::vector<double>>& mats, int S)
    {
        const int R = range.root_max - range.root_min + 1;
        const int nl = tree.n_leaves();
        const int trials = num_random_samples;
        std::vector<int> prefix;
        {
            std::vector<int> st{tree.root};
            while (!st.empty()) {
                const int v = st.back();
                st.pop_back();
                prefix.push_back(v);
                if (tree.left[v] >= 0) {
                    st.push_back(tree.right[v]);
                    st.push_back(tree.left[v]);
                }
            }
        }
        std::vector<int32_t> counts((size_t)R * trials * nl), lo((size_t)R * trials), hi((size_t)R * trials), cm((size_t)R * trials);
        // The random numbers are drawn first, in the reference's global order (root size, trial, node in prefix
        // order); root sizes are then independent of each other (the running-minimum column limit is per root
        // size), so worker threads take whole root sizes -- the reference threads split them the same way
        // (cafe/conditional_distribution.cpp:88-108) -- and the result does not depend on the thread count.
        const bool show_times = opt_timing;
        auto t_phase = std::chrono::steady_clock::now();
        auto phase = [&](const char* what) {
            if (!show_times) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "null: %-44s %.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
            t_phase = now;
        };
        const size_t per_family = prefix.size() - 1;
        std::vector<int32_t> rnd((size_t)R * trials * per_family);
        rng.fill_raw(rnd.data(), rnd.size());
        phase("random numbers in the reference's order");
        CdfCache cdf;
        cdf.reset(mats, S);
        cdf.build_all();  // read-only afterwards
        phase("cumulative rows of the matrices");
        auto sample_root_size = [&](int si) {
            const int s = range.root_min + si;
            const int maxFamilySize = std::max(s, range.max);  // get_random_probabilities :20
            int rmax = range.max;
            std::vector<int> fs(tree.n);
            size_t row = (size_t)si * trials;
            const int32_t* r = rnd.data() + row * per_family;
            for (int t = 0; t < trials; ++t, ++row) {
                int mx = 0;
                fs[tree.root] = s;
                for (int v : prefix) {
                    if (v == tree.root) continue;
                    const int c = cdf.draw(v, fs[tree.parent[v]], GlibcRand::to_unit(*r++), maxFamilySize);
                    fs[v] = c;
                    if (mx < c) mx = c;
                }
                rmax = std::min(mx + std::max(50, mx / 5), rmax);  // :29, running MIN over the trials of this root size
                for (int j = 0; j < nl; ++j) counts[row * nl + j] = fs[2 * j];
                lo[row] = hi[row] = s;
                cm[row] = rmax;
            }
        };
        {
            const int workers = std::max(1, std::min<int>({(int)std::thread::hardware_concurrency(), 32, R}));
            std::atomic<int> next{0};
            std::vector<std::thread> pool;
            for (int w = 0; w < workers; ++w)
                pool.emplace_back([&] {
                    for (int si = next++; si < R; si = next++) sample_root_size(si);
                });
            for (auto& th : pool) th.join();
        }
        phase("inverse-CDF draws (threads by root size)");
        const auto t_sampled = std::chrono::steady_clock::now();
        std::vector<double> probs((size_t)R * trials);
        if (report_sharded()) {
            // the draws above follow the reference's global order on every rank; the likelihoods of the simulated
            // families are sharded by root size (as the reference's threads are, cafe/conditional_distribution.cpp:88-108)
            int r0, r1;
            block_of(R, shard_rank, r0, r1);
            const size_t b0 = (size_t)r0 * trials, nb = (size_t)(r1 - r0) * trials;
            if (nb)
                hip_check(cafehip_eval_root_likelihoods(ctx, (int)nb, counts.data() + b0 * nl, lo.data() + b0, hi.data() + b0,
                                                        cm.data() + b0, probs.data() + b0));
            auto all = gather_blocks(probs.data() + b0, [&](int r) {
                int a, b;
                block_of(R, r, a, b);
                return (long long)(b - a) * trials * (long long)sizeof(double);
            });
            memcpy(probs.data(), all.data(), all.size());
        } else {
            hip_check(cafehip_eval_root_likelihoods(ctx, R * trials, counts.data(), lo.data(), hi.data(), cm.data(), probs.data()));
        }
        if (show_times)
            fprintf(stderr, "null: likelihoods of %d simulated families %.3f s (GPU)\n", R * trials,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_sampled).count());
        cond_dist.assign(R, std::vector<double>(trials));
        for (int i = 0; i < R; ++i) {
            std::copy(probs.begin() + (size_t)i * trials, probs.begin() + (size_t)(i + 1) * trials, cond_dist[i].begin());
            std::sort(cond_dist[i].begin(), cond_dist[i].end());  // :41
        }
        phase("likelihoods (GPU) + sorting the samples");
    }

    int cmd_report(const std::vector<std::string>& tokens)
    {  // cafe_cmd_report / cafe_do_report, cafe/cafe_commands.cpp:1010-1018, cafe/reports.cpp:650-708 (text format)
        prereqs(true, true);
        require_single_model("report");
        const bool show_times = opt_timing;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!show_times) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "report: %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
            t_last = now;
        };
        if ((int)params.size() != num_params || num_params == 0)
            throw std::runtime_error("ERROR: Lambda values were not set. Please set lambda values with the 'lambda' or 'lambdamu' command.\n");
        if (tokens.size() < 2) throw std::runtime_error("Usage(report): report <name>");
        // get_report_parameters, cafe/reports.cpp:599-628: `save` writes the report of the state already computed (no
        // Monte-Carlo null, no Viterbi pass: cafe_do_report's just_save path, :650-708) -- the reference's only report
        // checkpoint; html / json / branchcutting / likelihood / lh2 are formats and tests outside SURVEY.md section 8
        bool just_save = false;
        for (size_t i = 2; i < tokens.size(); ++i) {
            if (iequals(tokens[i], "save")) {
                just_save = true;
                break;
            }
            throw std::runtime_error("report " + tokens[i] + " is outside this build's scope (text report and `save` only; html, json, "
                                     "branchcutting, likelihood and lh2 are SURVEY.md section 2 OUT OF SCOPE)");
        }
        const std::string name = tokens[1];
        ReportArith arith(this);
        if (just_save) {
            if (rep_sizes.size() != (size_t)fam.F() * tree.n || rep_max_p.size() != (size_t)fam.F() || rep_expand.size() != (size_t)tree.n - 1)
                throw std::runtime_error("ERROR(report): nothing to save -- run `report <name>` on this table first");
            return write_report_text(name);
        }
        upload();
        // matrices for the fitted parameters, resident on the device + host copies for sampling and
        // for the exact ==/< comparisons of viterbi_sum_probabilities
        std::vector<double> nl_, nm_;
        node_rates(params.data(), nl_, nm_);
        reset_cache_exact(nl_, nm_);
        const int S = cafehip_matrix_size(ctx);
        std::vector<std::vector<double>> mats(tree.n);
        for (int v = 0; v < tree.n; ++v) {
            if (v == tree.root) continue;
            mats[v].resize((size_t)S * S);
            int s_out = 0;
            hip_check(cafehip_get_matrix(ctx, v, mats[v].data(), &s_out));
        }
        lap("matrices to host");
        if (cond_dist.empty()) compute_conditional_distribution(mats, S);
        lap("Monte-Carlo null");

        log("Running Viterbi algorithm....\n");
        const int F = fam.F(), nl = tree.n_leaves(), ns = (int)fam.species.size(), n = tree.n;
        // per-family ranges: cafe_family_set_size_with_family_forced, cafe/cafe_family.c:236-255
        std::vector<int32_t> counts((size_t)std::max(F, 1) * nl, 0), lo(F), hi(F), cm(F);
        std::vector<int64_t> off(F + 1, 0);
        for (int i = 0; i < F; ++i) {
            int mx = 0;
            for (int s = 0; s < ns; ++s) {
                if (species_index[s] < 0) continue;
                const int c = fam.counts[(size_t)i * ns + s];
                counts[(size_t)i * nl + species_index[s] / 2] = c;
                mx = std::max(mx, c);
            }
            lo[i] = 1;
            hi[i] = (int)std::rint(mx * 1.25);
            cm[i] = mx + std::max(50, mx / 5);
            off[i + 1] = off[i] + std::max(hi[i] - lo[i] + 1, 0);
        }
        // the device call wants non-empty root ranges: families with rfsize == 0 are scored on [1,1] and ignored
        std::vector<int32_t> hi_call(hi);
        std::vector<int64_t> off_call(F + 1, 0);
        for (int i = 0; i < F; ++i) {
            if (hi_call[i] < lo[i]) hi_call[i] = lo[i];
            off_call[i + 1] = off_call[i] + (hi_call[i] - lo[i] + 1);
        }
        std::vector<double> like((size_t)std::max<int64_t>(off_call[F], 1));
        rep_sizes.assign((size_t)F * n, 0);
        if (F && report_sharded()) {
            // each rank scores and back-tracks a contiguous block of the families; two gathers rebuild the full arrays
            int f0, f1;
            block_of(F, shard_rank, f0, f1);
            if (f1 > f0) {
                hip_check(cafehip_eval_root_likelihoods(ctx, f1 - f0, counts.data() + (size_t)f0 * nl, lo.data() + f0,
                                                        hi_call.data() + f0, cm.data() + f0, like.data() + off_call[f0]));
                hip_check(cafehip_viterbi(ctx, f1 - f0, counts.data() + (size_t)f0 * nl, lo.data() + f0, hi.data() + f0,
                                          cm.data() + f0, rep_sizes.data() + (size_t)f0 * n));
            }
            auto all_like = gather_blocks(like.data() + off_call[f0], [&](int r) {
                int a, b;
                block_of(F, r, a, b);
                return (long long)(off_call[b] - off_call[a]) * (long long)sizeof(double);
            });
            memcpy(like.data(), all_like.data(), all_like.size());
            auto all_sizes = gather_blocks(rep_sizes.data() + (size_t)f0 * n, [&](int r) {
                int a, b;
                block_of(F, r, a, b);
                return (long long)(b - a) * n * (long long)sizeof(int32_t);
            });
            memcpy(rep_sizes.data(), all_sizes.data(), all_sizes.size());
        } else if (F) {
            hip_check(cafehip_eval_root_likelihoods(ctx, F, counts.data(), lo.data(), hi_call.data(), cm.data(), like.data()));
            hip_check(cafehip_viterbi(ctx, F, counts.data(), lo.data(), hi.data(), cm.data(), rep_sizes.data()));
        }

        lap("root likelihoods + Viterbi");
        rep_max_p.assign(F, 0.0);
        rep_branch_p.assign((size_t)F * (n - 1), -1.0);
        const int npairs = n - 1;
        std::vector<double>& avg_exp = rep_avg_exp;      // kept in the session: `report <name> save` writes them again
        std::vector<int>&n_expand = rep_expand, &n_remain = rep_remain, &n_decrease = rep_decrease;
        avg_exp.assign(npairs, 0.0);
        n_expand.assign(npairs, 0);
        n_remain.assign(npairs, 0);
        n_decrease.assign(npairs, 0);
        // families are independent here; worker threads take contiguous blocks and keep private integer tallies
        // (exact in any order), so the report does not depend on the thread count
        struct Tally {
            std::vector<long long> delta;
            std::vector<int> expand, remain, decrease;
        };
        auto work = [&](int i0, int i1, Tally& t) {
            t.delta.assign(npairs, 0);
            t.expand.assign(npairs, 0);
            t.remain.assign(npairs, 0);
            t.decrease.assign(npairs, 0);
            for (int i = i0; i < i1; ++i) {
            // cafe_tree_p_values (cafe/pvalue.cpp:143-154) + viterbi_set_max_pvalue (cafe/viterbi.cpp:32-39)
            const int rf = hi[i] - lo[i] + 1;
            double maxp = 0;
            for (int s = 0; s < rf; ++s) {
                const double pv = pvalue_rank(like[off_call[i] + s], cond_dist[s].data(), num_random_samples);
                if (s == 0 || pv > maxp) maxp = pv;
            }
            rep_max_p[i] = maxp;
            const int32_t* fs = &rep_sizes[(size_t)i * n];
            // compute_size_deltas, cafe/viterbi.cpp:570-595
            for (int j = 0; j < (n - 1) / 2; ++j) {
                const int node = 2 * j + 1;
                const int child[2] = {tree.left[node], tree.right[node]};
                for (int k = 0; k < 2; ++k) {
                    const int m = 2 * j + k;
                    if (fs[child[k]] > fs[node]) t.expand[m]++;
                    else if (fs[child[k]] == fs[node]) t.remain[m]++;
                    else t.decrease[m]++;
                    t.delta[m] += fs[child[k]] - fs[node];
                }
            }
            if (maxp > pvalue) continue;  // cafe/viterbi.cpp:105-114: branch p-values stay -1
            // viterbi_sum_probabilities, cafe/viterbi.cpp:44-71
            for (int j = 0; j < (n - 1) / 2; ++j) {
                const int node = 2 * j + 1;
                const int child[2] = {tree.left[node], tree.right[node]};
                for (int k = 0; k < 2; ++k) {
                    const double* m = mats[child[k]].data() + (size_t)fs[node] * S;
                    const double p = m[fs[child[k]]];
                    double acc = 0;
                    for (int mm = 0; mm <= cm[i]; ++mm) {
                        if (m[mm] == p) acc += m[mm] / 2.0;
                        else if (m[mm] < p) acc += m[mm];
                    }
                    rep_branch_p[(size_t)i * (n - 1) + 2 * j + k] = acc;
                }
            }
            }
        };
        {
            const int workers = std::max(1, std::min<int>({(int)std::thread::hardware_concurrency(), 32, (F + 1023) / 1024}));
            std::vector<Tally> tallies(workers);
            std::vector<std::thread> pool;
            for (int w = 0; w < workers; ++w)
                pool.emplace_back([&, w] { work((int)((long long)F * w / workers), (int)((long long)F * (w + 1) / workers), tallies[w]); });
            for (auto& th : pool) th.join();
            std::vector<long long> delta(npairs, 0);
            for (auto& t : tallies)
                for (int m = 0; m < npairs; ++m) {
                    delta[m] += t.delta[m];
                    n_expand[m] += t.expand[m];
                    n_remain[m] += t.remain[m];
                    n_decrease[m] += t.decrease[m];
                }
            for (int m = 0; m < npairs; ++m) avg_exp[m] = (double)delta[m];
        }
        for (double& v : avg_exp) v /= std::max(F, 1);
        lap("p-values (host)");
        const int rc = write_report_text(name);
        lap("writing the file");
        return rc;
    }

    // ---- text report: operator<<(ostream&, const Report&), cafe/reports.cpp:453-501 ----
    int write_report_text(const std::string& name)
    {
        const int F = fam.F(), n = tree.n, npairs = n - 1;
        const std::vector<double>& avg_exp = rep_avg_exp;
        const std::vector<int>&n_expand = rep_expand, &n_remain = rep_remain, &n_decrease = rep_decrease;
        if (shard_world > 1 && shard_rank != 0) {  // one writer
            log("Report Done\n");
            return 0;
        }
        const std::string filename = name + ".cafe";
        FILE* fp = fopen(filename.c_str(), "w");
        if (!fp) throw std::runtime_error("ERROR(report) : Cannot open " + name + " in write mode.\n");
        fprintf(fp, "Tree:%s\n", tree_string([&](int v) { return tree.name[v]; }, true).c_str());
        fprintf(fp, "Lambda:");
        for (int i = 0; i < num_lambdas; ++i) fprintf(fp, "\t%s", fmt_g(params[i]).c_str());
        fprintf(fp, "\n");
        if (have_lambda_tree)
            fprintf(fp, "Lambda tree:\t%s\n", tree_string([&](int v) { return lambda_tree.name[v]; }, false).c_str());
        fprintf(fp, "# IDs of nodes:%s\n",
                tree_string([&](int v) { return tree.name[v] + "<" + std::to_string(v) + ">"; }, false).c_str());
        fprintf(fp, "# Output format for: ' Average Expansion', 'Expansions', 'No Change', 'Contractions', and "
                    "'Branch-specific P-values' = (node ID, node ID): ");
        for (int b = 1; b < n; b += 2) fprintf(fp, "(%d,%d) ", tree.left[b], tree.right[b]);
        fprintf(fp, "\n# Output format for 'Branch cutting P-values' and 'Likelihood Ratio Test': (0");
        for (int i = 1; i < n; ++i) fprintf(fp, ", %d", i);
        fprintf(fp, ")\n");
        fprintf(fp, "Average Expansion:");
        for (int b = 0; b < npairs / 2; ++b) fprintf(fp, "\t(%s,%s)", fmt_g(avg_exp[2 * b]).c_str(), fmt_g(avg_exp[2 * b + 1]).c_str());
        fprintf(fp, "\nExpansion :");
        for (int b = 0; b < npairs / 2; ++b) fprintf(fp, "\t(%d,%d)", n_expand[2 * b], n_expand[2 * b + 1]);
        fprintf(fp, "\nnRemain :");
        for (int b = 0; b < npairs / 2; ++b) fprintf(fp, "\t(%d,%d)", n_remain[2 * b], n_remain[2 * b + 1]);
        fprintf(fp, "\nnDecrease :");
        for (int b = 0; b < npairs / 2; ++b) fprintf(fp, "\t(%d,%d)", n_decrease[2 * b], n_decrease[2 * b + 1]);
        fprintf(fp, "\n'ID'\t'Newick'\t'Family-wide P-value'\t'Viterbi P-values'\t'cut P-value'\t'Likelihood Ratio'\n");
        // The Newick text of a family is the tree's text with "_<size>" after every node name: the constant
        // pieces are cut once (sentinel sizes mark the holes), each row then only formats its integers.
        std::vector<std::string> pieces;
        std::vector<int> hole_node;
        {
            const std::string marked = tree_string([&](int v) { return tree.name[v] + "_\x01" + std::to_string(v) + "\x02"; }, true);
            size_t pos = 0;
            while (true) {
                const size_t a = marked.find('\x01', pos);
                if (a == std::string::npos) {
                    pieces.push_back(marked.substr(pos));
                    break;
                }
                const size_t b = marked.find('\x02', a);
                pieces.push_back(marked.substr(pos, a - pos));
                hole_node.push_back(atoi(marked.substr(a + 1, b - a - 1).c_str()));
                pos = b + 1;
            }
        }
        // rows are formatted by worker threads over contiguous blocks of families and written in family order
        auto format_rows = [&](int i0, int i1, std::string& out) {
            out.reserve((size_t)(i1 - i0) * (pieces.size() * 8 + (size_t)npairs * 12 + 64));
            for (int i = i0; i < i1; ++i) {
                const int32_t* fs = &rep_sizes[(size_t)i * n];
                out += fam.ids[i];
                out += '\t';
                for (size_t h = 0; h < hole_node.size(); ++h) {
                    out += pieces[h];
                    append_int(out, fs[hole_node[h]]);
                }
                out += pieces.back();
                out += '\t';
                append_g(out, rep_max_p[i]);
                out += "\t(";
                for (int b = 0; b < npairs / 2; ++b) {
                    const double p1 = rep_branch_p[(size_t)i * (n - 1) + 2 * b], p2 = rep_branch_p[(size_t)i * (n - 1) + 2 * b + 1];
                    if (p1 < 0) {
                        out += "(-,-)";
                    } else {
                        out += '(';
                        append_g(out, p1);
                        out += ',';
                        append_g(out, p2);
                        out += ')';
                    }
                    if (b < npairs / 2 - 1) out += ',';
                }
                out += ")\t\n";
            }
        };
        {
            const int workers = std::max(1, std::min<int>({(int)std::thread::hardware_concurrency(), 32, (F + 2047) / 2048}));
            std::vector<std::string> chunks(workers);
            std::vector<std::thread> pool;
            for (int w = 0; w < workers; ++w)
                pool.emplace_back([&, w] { format_rows((int)((long long)F * w / workers), (int)((long long)F * (w + 1) / workers), chunks[w]); });
            for (auto& th : pool) th.join();
            for (auto& c : chunks) fwrite(c.data(), 1, c.size(), fp);
        }
        fclose(fp);
        log("Report Done\n");
        return 0;
    }


    // ---- errormodel: cafe_cmd_errormodel (cafe/cafe_commands.cpp:1608-1639),
    //      set_error_matrix_from_file (cafe/error_model.cpp:231-259), file reader (:145-204),
    //      __check_error_model_columnsums (cafe/cafe_shell.c:585-622) ----
    void read_error_model(const std::string& file)
    {
        std::ifstream in(file);
        if (!in) throw std::runtime_error("ERROR(errormodel): Cannot open " + file + " in read mode.");
        std::string line;
        if (!std::getline(in, line)) throw std::runtime_error("Empty file");
        auto data = HostFamilies::split(line, ' ');
        auto mx = HostFamilies::split(data.at(0), ':');
        if (mx.size() < 2) throw std::runtime_error("errormodel: first line must be maxcnt:<n>");
        const int file_row_count = atoi(mx[1].c_str());
        int mfs = std::max(range.max, file_row_count);  // :154-155, :241
        int fromdiff = 0, todiff = 0;
        if (std::getline(in, line)) {
            data = HostFamilies::split(line, ' ');
            fromdiff = atoi(data.at(1).c_str());
            todiff = atoi(data.at(data.size() - 1).c_str());
        }
        const int ld = mfs + 1;
        std::vector<double> E((size_t)ld * ld, 0.0);
        int j = 0;
        while (std::getline(in, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            data = HostFamilies::split(line, ' ');
            if ((int)data.size() != (todiff - fromdiff) + 2) continue;
            const int col1 = atoi(data[0].c_str());
            if (col1 != j)
                throw std::runtime_error("errormodel: rows must be consecutive (row " + std::to_string(j) + " expected, found " +
                                         std::to_string(col1) + "); the reference loops forever on such files (cafe/error_model.cpp:171-180)");
            int k = 1;
            for (int i = fromdiff; i <= todiff; ++i, ++k)
                if (i + j >= 0 && i + j <= mfs) E[(size_t)(i + j) * ld + j] = atof(data[k].c_str());
            ++j;
        }
        while (j && j <= mfs) {  // :193-201 copy the previous row's band down the diagonal
            for (int i = fromdiff; i <= todiff; ++i)
                if (i + j >= 0 && i + j <= mfs) E[(size_t)(i + j) * ld + j] = E[(size_t)(i + j - 1) * ld + (j - 1)];
            ++j;
        }
        // __check_error_model_columnsums: NB the reference calls the INTEGER abs() on a double there, so the
        // middle columns are renormalised only when |1 - sum| >= 1 (cafe/cafe_shell.c:603)
        const int diff = todiff;
        auto colsum = [&](int c) {
            double sum = 0;
            for (int i = 0; i <= mfs; ++i) sum += E[(size_t)i * ld + c];
            return sum;
        };
        for (int c = 0; c < diff && c <= mfs; ++c) E[(size_t)0 * ld + c] += (1 - colsum(c));
        for (int c = diff; c <= mfs - diff; ++c) {
            const double sum = colsum(c);
            if (std::abs((int)(1 - sum)) > 0.00000000000001)
                for (int i = 0; i <= mfs; ++i) E[(size_t)i * ld + c] /= sum;
        }
        for (int c = std::max(mfs - diff + 1, 0); c <= mfs; ++c) E[(size_t)mfs * ld + c] += (1 - colsum(c));
        err_matrix.swap(E);
        err_mfs = mfs;
        err_fromdiff = fromdiff;
        err_todiff = todiff;
        err_file = file;
    }

    int cmd_errormodel(const std::vector<std::string>& tokens)
    {
        prereqs(true, true);
        auto args = build_argument_list(tokens);
        std::string model;
        std::vector<std::string> species;
        bool all = false;
        for (auto& a : args) {
            if (a.opt == "-model" && !a.argv.empty()) model = a.argv[0];
            else if (a.opt == "-sp") species.insert(species.end(), a.argv.begin(), a.argv.end());
            else if (a.opt == "-all") all = true;
            else throw std::runtime_error("errormodel " + a.opt + " is outside this build's scope (supported: -model -sp -all)");
        }
        if (model.empty()) throw std::runtime_error("ERROR(errormodel): we need an error model specified (-model) or two data files.\n");
        if (!err_file.empty() && !iequals(err_file, model))
            throw std::runtime_error("errormodel: one model file per session is supported (already using " + err_file + ")");
        if (err_file.empty()) {
            read_error_model(model);
            err_leaf.assign(tree.n, 0);
        }
        // init_error_ptr, cafe/error_model.cpp:206-229
        if (!species.empty()) {
            for (auto& spname : species)
                for (size_t s_ = 0; s_ < fam.species.size(); ++s_)
                    if (iequals(fam.species[s_], spname) && species_index[s_] >= 0) err_leaf[species_index[s_]] = 1;
        } else if (all) {
            for (size_t s_ = 0; s_ < fam.species.size(); ++s_)
                if (species_index[s_] >= 0) err_leaf[species_index[s_]] = 1;
        }
        device_families_current = false;
        spec.clear();
        fprintf(stderr, "errormodel: %s set.\n", model.c_str());
        return 0;
    }


    // ---- helpers shared by genfamily / rootdist / pvalue / lhtest ----
    std::vector<int> prefix_order() const
    {
        std::vector<int> prefix, st{tree.root};
        while (!st.empty()) {
            const int v = st.back();
            st.pop_back();
            prefix.push_back(v);
            if (tree.left[v] >= 0) {
                st.push_back(tree.right[v]);
                st.push_back(tree.left[v]);
            }
        }
        return prefix;
    }

    // Matrices of the report phase (Monte-Carlo null, Viterbi, branch p-values, genfamily, rootdist) are built in
    // the reference's per-term arithmetic: random draws are compared with cumulative sums of matrix rows
    // (cafe/cafe_tree.c:533-569) and viterbi_sum_probabilities uses exact == / < on entries
    // (cafe/viterbi.cpp:60-67), so a last-bit difference can flip a decision; the build is a one-off here.
    void reset_cache_exact(const std::vector<double>& nl_, const std::vector<double>& nm_)
    {
        hip_check(cafehip_set_exact_matrices(ctx, 1));
        const int rc = cafehip_reset_birthdeath_cache(ctx, nl_.data(), nm_.data());
        cafehip_set_exact_matrices(ctx, 0);
        hip_check(rc);
    }

    // matrices for the current parameters on the device + host copies (node -> S x S)
    int fetch_matrices(std::vector<std::vector<double>>& mats)
    {
        if ((int)params.size() != num_params || num_params == 0)
            throw std::runtime_error("ERROR: Lambda values were not set. Please set lambda values with the 'lambda' or 'lambdamu' command.\n");
        upload();
        std::vector<double> nl_, nm_;
        node_rates(params.data(), nl_, nm_);
        reset_cache_exact(nl_, nm_);
        const int S = cafehip_matrix_size(ctx);
        mats.assign(tree.n, {});
        for (int v = 0; v < tree.n; ++v) {
            if (v == tree.root) continue;
            mats[v].resize((size_t)S * S);
            int s_out = 0;
            hip_check(cafehip_get_matrix(ctx, v, mats[v].data(), &s_out));
        }
        return S;
    }

    // cafe_tree_random_familysize, cafe/cafe_tree.c:533-569
    int random_familysize(CdfCache& cdf, const std::vector<int>& prefix, int root_size, int max_family_size,
                          std::vector<int>& fs)
    {
        int mx = 0;
        fs[tree.root] = root_size;
        for (int v : prefix) {
            if (v == tree.root) continue;
            const double rnd = unifrnd();
            const int c = cdf.draw(v, fs[tree.parent[v]], rnd, max_family_size);
            fs[v] = c;
            if (mx < c) mx = c;
        }
        return mx;
    }

    // get_root_dist, cafe/cafe_commands.cpp:619-646: Viterbi root size of every family under the GLOBAL ranges
    void compute_root_dist()
    {
        std::vector<std::vector<double>> mats;
        fetch_matrices(mats);
        const int F = fam.F(), nl = tree.n_leaves(), ns = (int)fam.species.size(), n = tree.n;
        std::vector<int32_t> counts((size_t)std::max(F, 1) * nl, 0), lo(F, range.root_min), hi(F, range.root_max), cm(F, range.max);
        for (int i = 0; i < F; ++i)
            for (int s_ = 0; s_ < ns; ++s_)
                if (species_index[s_] >= 0) counts[(size_t)i * nl + species_index[s_] / 2] = fam.counts[(size_t)i * ns + s_];
        std::vector<int32_t> sizes((size_t)std::max(F, 1) * n);
        if (shard_rank == 0) {
            printf("Viterbi\n");
            fflush(stdout);
        }
        if (F) hip_check(cafehip_viterbi(ctx, F, counts.data(), lo.data(), hi.data(), cm.data(), sizes.data()));
        root_dist.assign(range.root_max - range.root_min + 2, 0);
        for (int i = 0; i < F; ++i) {
            const int rs = sizes[(size_t)i * n + tree.root];
            if (rs >= 0 && rs < (int)root_dist.size()) root_dist[rs]++;
        }
    }

    // cafe_cmd_score + cafe_shell_score, cafe/cafe_commands.cpp:2124-2209: the objective at the current parameters
    // (the objective logs its own line, as the reference's search function does), the summary line again, the
    // score through `ostream << double`
    int cmd_score(const std::vector<std::string>&)
    {
        prereqs(true, true);
        if (params.empty()) throw std::runtime_error("ERROR: Lambda values were not set. Please set lambda values with the 'lambda' or 'lambdamu' command.\n");
        double score;
        if (k_clusters > 0) {
            score = -cluster_objective(params.data());
            const int fix = fixcluster0 ? 1 : 0;
            log("Lambda : %s%s\n", fix ? "0," : "", join_double(params.data(), num_lambdas * (k_clusters - fix)).c_str());
            log("p : %s\n", join_double(k_weights.data(), k_clusters).c_str());
            log("Score: %f\n", score);
        } else {
            score = -objective(params.data());
            if (has_mu) {
                log("Lambda : %s ", join_double(params.data(), num_lambdas).c_str());
                log("Mu : %s & Score: %f\n", join_double(params.data() + num_lambdas, num_mus).c_str(), score);
            } else {
                log("Lambda : %s & Score: %f\n", join_double(params.data(), num_lambdas).c_str(), score);
            }
        }
        std::ostringstream os;
        os << score << std::endl;
        log("%s", os.str().c_str());
        if (k_clusters > 0) log_cluster_membership();
        last_score = -score;
        return 0;
    }

    int cmd_rootdist(const std::vector<std::string>& tokens)
    {  // cafe_cmd_rootdist, cafe/cafe_commands.cpp:1831-1916
        prereqs(false, true);
        auto args = build_argument_list(tokens);
        std::string file;
        for (auto& a : args)
            if (a.opt == "-i" && !a.argv.empty()) file = a.argv[0];
        if (tokens.size() < 2) {
            prereqs(true, false);
            compute_root_dist();
            log("-----------------------------------------------------------\n");
            log("Family information: %s\n", fam.path.c_str());
            log("Log: %s\n", log_name.c_str());
            log("Tree: %s\n", tree_string([&](int v) { return tree.name[v]; }, true).c_str());
            log("The number of families is %d\n", fam.F());
            return 0;
        }
        if (file.empty()) throw std::runtime_error("Usage(rootdist): rootdist [-i file]");
        std::ifstream in(file);
        if (!in) throw std::runtime_error("ERROR(rootdist): Cannot open " + file + " in read mode.");
        std::string line;
        if (!std::getline(in, line)) throw std::runtime_error("Empty file: " + file);
        auto data = HostFamilies::split(line, ' ');
        auto mx = HostFamilies::split(data.back(), ':');
        if (mx.size() < 2) throw std::runtime_error("Invalid format in rootdist file");
        const int max_rootsize = atoi(mx[1].c_str());
        root_dist.assign(max_rootsize + 1, 0);
        range.root_min = 1;  // :1889-1893
        range.root_max = max_rootsize;
        range.min = 0;
        range.max = max_rootsize * 2;
        device_families_current = false;
        while (std::getline(in, line)) {
            data = HostFamilies::split(line, ' ');
            if (data.size() < 2) continue;
            const int idx = atoi(data[0].c_str());
            if (idx >= 0 && idx <= max_rootsize) root_dist[idx] = atoi(data[1].c_str());
        }
        return 0;
    }

    int cmd_genfamily(const std::vector<std::string>& tokens)
    {  // cafe_cmd_generate_random_family, cafe/cafe_commands.cpp:718-815
        if (tokens.size() == 1) throw std::runtime_error("Usage: genfamily directory/fileprefix -t integer");
        prereqs(false, true);
        require_single_model("genfamily");
        int num_trials = 1;
        for (auto& a : build_argument_list(tokens))
            if (a.opt == "-t" && !a.argv.empty()) num_trials = atoi(a.argv[0].c_str());
        const std::string prefix_path = tokens[1];
        if (root_dist.empty()) {
            prereqs(true, false);
            compute_root_dist();
        } else {
            log("Using user defined root size distribution for simulation... \n");
            if (!have_family) {
                // no table loaded: the device only needs the ranges, give it an empty table
                fam = HostFamilies();
                for (int i = 0; i < tree.n; i += 2) fam.species.push_back(tree.name[i]);
                have_family = true;
                sync_species_index();
            }
        }
        std::vector<std::vector<double>> mats;
        const int S = fetch_matrices(mats);
        const std::vector<int> prefix = prefix_order();
        const int rfsize = range.root_max - range.root_min + 1;
        const int maxFamilysize = S - 1;  // probability_cache->maxFamilysize
        std::vector<int> fs(tree.n, 0);
        CdfCache cdf;
        cdf.reset(mats, S);
        for (int t = 0; t < num_trials; ++t) {
            const std::string base = prefix_path + "_" + std::to_string(t + 1);
            // sharded job: every rank simulates (the random stream must stay in step), ONE rank writes
            const bool writer = shard_rank == 0;
            FILE* ft = fopen(writer ? (base + ".tab").c_str() : "/dev/null", "w");
            if (!ft) throw std::runtime_error(base + ".tab failed to open");
            FILE* fr = fopen(writer ? (base + ".truth").c_str() : "/dev/null", "w");
            if (!fr) {
                fclose(ft);
                throw std::runtime_error(base + ".truth failed to open");
            }
            // write_node_headers :672-694
            fprintf(ft, "DESC\tFID");
            fprintf(fr, "DESC\tFID");
            for (int i = 0; i < tree.n; i += 2) fprintf(ft, "\t%s", tree.name[i].c_str());
            fprintf(ft, "\n");
            for (int i = 0; i < tree.n; ++i) {
                if (!tree.name[i].empty()) fprintf(fr, "\t%s", tree.name[i].c_str());
                else fprintf(fr, "\t-%d", i);
            }
            fprintf(fr, "\n");
            int id = 1;
            for (int i = 1; i <= rfsize && i < (int)root_dist.size(); ++i) {
                for (int j = 0; j < root_dist[i]; ++j) {
                    random_familysize(cdf, prefix, i, maxFamilysize, fs);
                    fprintf(ft, "root%d\t%d", i, id);  // write_leaves :696-711
                    fprintf(fr, "root%d\t%d", i, id);
                    for (int n_ = 0; n_ < tree.n; n_ += 2) fprintf(ft, "\t%d", fs[n_]);
                    for (int n_ = 0; n_ < tree.n; ++n_) fprintf(fr, "\t%d", fs[n_]);
                    fprintf(ft, "\n");
                    fprintf(fr, "\n");
                    ++id;
                }
            }
            fclose(ft);
            fclose(fr);
        }
        if (report_sharded()) {
            // nobody goes on (e.g. to an lhtest over these files) before the writer has closed them
            char token = 0;
            std::vector<char> every(shard_world);
            if (allgather(allgather_user, &token, 1, every.data(), 1) != 0) throw std::runtime_error("genfamily: barrier failed");
        }
        return 0;
    }

    // get_posterior for the current parameters and a given prior (cafe/lambda.cpp:691-724), -inf on a zero family
    double posterior_with_prior(const std::vector<double>& pr)
    {
        if (shard_world > 1 && !native_comm)
            throw std::runtime_error("lhtest under a callback exchange: the caller's buffers are sized for the table of the "
                                     "last `lambda`; use the native communicator (cafehost_init_comm) or one rank");
        upload();
        std::vector<double> nl_, nm_;
        node_rates(params.data(), nl_, nm_);
        int32_t zero = -1;
        double score = evaluate(nl_, nm_, pr, zero);
        if (zero >= 0) {
            fprintf(stderr, "WARNING: Calculated posterior probability for family %s = 0\n", fam.ids[zero].c_str());
            score = std::log(0.0);
        }
        return score;
    }

    int cmd_lhtest(const std::vector<std::string>& tokens)
    {  // cafe_cmd_lhtest, cafe/cafe_commands.cpp:1473-1536
        prereqs(false, true);
        require_single_model("lhtest");
        if (shard_world > 1 && !native_comm)
            throw std::runtime_error("lhtest under a callback exchange is not supported: every file it loads needs the "
                                     "exchange re-wired; use the native communicator (cafehost_init_comm) or one rank");
        std::string dir, ltree, outfile;
        double lam = 0.0;
        for (auto& a : build_argument_list(tokens)) {
            if (a.argv.empty()) continue;
            if (a.opt == "-d") dir = a.argv[0];
            if (a.opt == "-l") lam = atof(a.argv[0].c_str());
            if (a.opt == "-t") ltree = a.argv[0];
            if (a.opt == "-o") outfile = a.argv[0];
        }
        if (dir.empty() || ltree.empty()) throw std::runtime_error("Usage(lhtest): lhtest -d directory -l lambda -t lambdatree [-o outfile]");
        FILE* fout = stdout;
        const bool writer = shard_rank == 0;   // sharded: every rank runs the searches (identical decisions), one writes
        if (!writer) outfile = "/dev/null";
        if (!outfile.empty() && !(fout = fopen(outfile.c_str(), "w"))) throw std::runtime_error("Failed to open output file");
        std::vector<std::string> files;
        {
            DIR* d = opendir(dir.c_str());  // cafe/cafe_commands.cpp:1483-1490 walks the directory the same way
            if (!d) throw std::runtime_error("Failed to read directory " + dir);
            while (struct dirent* e = readdir(d)) {
                const std::string f = e->d_name;
                if (f.size() > 4 && f.substr(f.size() - 4) == ".tab" && f[0] != '.') files.push_back(f);
            }
            closedir(d);
        }
        std::sort(files.begin(), files.end());  // the reference uses readdir order; sorted here for reproducibility
        const std::string tree_str = tree_string([&](int v) { return tree.name[v]; }, true);
        std::vector<double> pr = prior;  // the prior in force when lhtest starts is used for every file (:1492-1493)
        if (pr.size() < 1000) pr.assign(1000, 0.0);
        char lbuf[64];
        snprintf(lbuf, sizeof lbuf, "%lf", lam);
        // the two searches of one file (:1497-1527) -> its output line
        auto one_file = [&](const std::string& f) {
            std::string line;
            char b[64];
            dispatch("load -i " + dir + "/" + f + " -p 0.01 -t 10 -l " + log_name);
            dispatch("tree " + tree_str);
            dispatch(std::string("lambda -s -l ") + lbuf);
            snprintf(b, sizeof b, "\t%lf\t%lf", posterior_with_prior(pr), params[0]);
            line += b;
            dispatch(std::string("lambda -s -v ") + lbuf + " -t " + ltree);
            snprintf(b, sizeof b, "\t%lf", posterior_with_prior(pr));
            line += b;
            for (int j = 0; j < num_lambdas; ++j) {
                snprintf(b, sizeof b, "\t%lf", params[j]);
                line += b;
            }
            return line + "\n";
        };
        const bool deal = native_comm && shard_world > 1 && opt_lhtest_deal != 0 && !files.empty();
        const auto t_lh0 = std::chrono::steady_clock::now();
        auto report_time = [&](const char* how) {
            if (opt_timing && shard_rank == 0)
                fprintf(stderr, "lhtest: %zu files in %.4f s (%s, %d rank%s)\n", files.size(),
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_lh0).count(), how, shard_world, shard_world > 1 ? "s" : "");
        };
        if (!deal) {
            // one rank -- or every rank running every search on its block of every table (option lhtest_deal=0)
            for (auto& f : files) {
                const std::string line = one_file(f);
                fputs(line.c_str(), fout);
                fflush(fout);
            }
            if (fout != stdout) fclose(fout);
            report_time(shard_world > 1 ? "every rank runs every file on its block" : "one rank");
            return 0;
        }
        // Sharded job: the files are independent full searches on small tables ("an outer embarrassingly-
        // parallel axis across GPUs").  Cutting a 59-family table N ways buys nothing and costs an exchange per evaluation;
        // instead rank r takes files r, r + N, ... and runs the WHOLE pipeline of each alone on its GPU (`solo`), the
        // formatted lines are gathered and rank 0 writes them in file order -- byte for byte the one-rank output.  The
        // random starts come from ONE stream in the one-rank run: a file consumes a number of draws that depends only on
        // the command shapes (prior fit 1 + one per lambda of each search), so a rank steps over the files it does not
        // run (checked against the count its own files really took).
        have_lambda_tree = false;
        set_lambda_tree(ltree);
        const unsigned long long draws_per_file = (1 + 1) + (1 + (unsigned long long)num_lambdas);
        std::string mine, problem;
        solo = true;
        device_families_current = false;
        try {
            for (size_t i = 0; i < files.size(); ++i) {
                if ((int)(i % shard_world) != shard_rank) {
                    rng.skip(draws_per_file);
                    continue;
                }
                const unsigned long long before = rng.draws;
                mine += one_file(files[i]);
                if (rng.draws - before != draws_per_file)
                    throw std::runtime_error("lhtest: a file took " + std::to_string(rng.draws - before) + " random draws where " +
                                             std::to_string(draws_per_file) + " were stepped over on the other ranks");
            }
        } catch (const std::exception& e) {
            problem = e.what();   // the others must hear of it: they are about to wait for this rank's lines
        }
        solo = false;
        device_families_current = false;
        // gather: [status byte][lines] per rank, sizes first
        std::string payload = problem.empty() ? std::string(1, 'k') + mine : std::string(1, 'E') + problem;
        long long my_size = (long long)payload.size();
        std::vector<long long> sizes(shard_world, 0);
        if (allgather(allgather_user, &my_size, sizeof my_size, sizes.data(), sizeof(long long)) != 0) throw std::runtime_error("lhtest: allgather failed");
        long long slot = 1;
        for (long long v : sizes) slot = std::max(slot, v);
        std::vector<char> all((size_t)slot * shard_world);
        if (allgather(allgather_user, payload.data(), my_size, all.data(), slot) != 0) throw std::runtime_error("lhtest: allgather failed");
        std::vector<std::string> lines_of(shard_world);
        for (int r = 0; r < shard_world; ++r) {
            const char* p = all.data() + (size_t)r * slot;
            if (sizes[r] < 1 || p[0] != 'k') {
                if (fout != stdout) fclose(fout);
                throw std::runtime_error("lhtest: rank " + std::to_string(r) + " failed: " + std::string(p + 1, p + std::max<long long>(sizes[r], 1)));
            }
            lines_of[r].assign(p + 1, p + sizes[r]);
        }
        std::vector<size_t> cursor(shard_world, 0);
        for (size_t i = 0; i < files.size(); ++i) {
            const int r = (int)(i % shard_world);
            const size_t e = lines_of[r].find('\n', cursor[r]);
            if (e == std::string::npos) throw std::runtime_error("lhtest: rank " + std::to_string(r) + " returned too few lines");
            fwrite(lines_of[r].data() + cursor[r], 1, e + 1 - cursor[r], fout);
            cursor[r] = e + 1;
        }
        fflush(fout);
        if (fout != stdout) fclose(fout);
        // Leave every rank where the one-rank run ends: the LAST file's table loaded (sharded again), its second
        // search's model and prior in force -- the owner of that file hands its results round.
        {
            const int owner = (int)((files.size() - 1) % shard_world);
            const int np = num_lambdas;
            std::vector<double> blob((size_t)np + 4, 0.0), every(((size_t)np + 4) * shard_world, 0.0);
            if (shard_rank == owner) {
                std::copy(params.begin(), params.begin() + std::min<size_t>(params.size(), np), blob.begin());
                blob[np] = poisson_lambda;
                blob[np + 1] = last_score;
                blob[np + 2] = search_iters;
                blob[np + 3] = n_evals;
            }
            if (allgather(allgather_user, blob.data(), (long long)(blob.size() * sizeof(double)), every.data(), (long long)(blob.size() * sizeof(double))) != 0)
                throw std::runtime_error("lhtest: allgather failed");
            const double* ob = every.data() + (size_t)owner * blob.size();
            mute_log = true;
            try {
                dispatch("load -i " + dir + "/" + files.back() + " -p 0.01 -t 10 -l " + log_name);
                dispatch("tree " + tree_str);
            } catch (...) {
                mute_log = false;
                throw;
            }
            mute_log = false;
            has_mu = false;
            eqbg = false;
            num_mus = 0;
            k_clusters = 0;
            have_lambda_tree = false;
            set_lambda_tree(ltree);
            num_params = num_lambdas;
            params.assign(ob, ob + np);
            poisson_lambda = ob[np];
            last_score = ob[np + 1];
            search_iters = (int)ob[np + 2];
            n_evals = (int)ob[np + 3];
            prior.assign(1000, 0.0);
            for (int i = 0; i < 1000; ++i) prior[i] = poisspdf(range.root_min - 1 + i, poisson_lambda);
            spec.clear();
            upload();
            std::vector<double> nl_, nm_;
            node_rates(params.data(), nl_, nm_);
            bool ok = true;
            for (double v : nl_) ok = ok && v >= 0;
            if (ok) hip_check(cafehip_reset_birthdeath_cache(ctx, nl_.data(), nm_.data()));
        }
        report_time("files dealt to the ranks");
        return 0;
    }

    int cmd_pvalue(const std::vector<std::string>& tokens)
    {  // cafe_cmd_pvalue, cafe/cafe_commands.cpp:1373-1412 (-o / -i / -idx)
        prereqs(false, true);
        require_single_model("pvalue");
        std::string outfile, infile;
        int index = -1;
        for (auto& a : build_argument_list(tokens)) {
            if (a.argv.empty()) continue;
            if (a.opt == "-o") outfile = a.argv[0];
            if (a.opt == "-i") infile = a.argv[0];
            if (a.opt == "-idx") index = atoi(a.argv[0].c_str());
        }
        ReportArith arith(this);
        if (!outfile.empty()) {
            std::vector<std::vector<double>> mats;
            const int S = fetch_matrices(mats);
            compute_conditional_distribution(mats, S);
            if (shard_world > 1 && shard_rank != 0) return 0;  // one writer
            FILE* fp = fopen(outfile.c_str(), "w");
            if (!fp) throw std::runtime_error("ERROR(pvalue): Cannot open " + outfile + " in write mode.");
            // write_pvalues, cafe/pvalue.cpp:63-77 (the reference passes a zeroed copy here, cafe_commands.cpp:1351-1361;
            // the distribution itself is written instead)
            for (auto& row : cond_dist) {
                for (size_t j = 0; j < row.size(); ++j) fprintf(fp, j ? "\t%.9g" : "%.9g", row[j]);
                fprintf(fp, "\n");
            }
            fclose(fp);
            return 0;
        }
        if (!infile.empty()) {
            log("Loading p-values ... \n");
            std::ifstream in(infile);
            if (!in) throw std::runtime_error("ERROR(pvalue): Cannot open " + infile + " in read mode.");
            cond_dist.clear();  // read_pvalues, cafe/pvalue.cpp:80-93
            std::string line;
            while (std::getline(in, line)) {
                std::vector<double> row(num_random_samples, 0.0);
                std::istringstream iss(line);
                for (int i = 0; i < num_random_samples; ++i) iss >> row[i];
                cond_dist.push_back(row);
            }
            log("Done Loading p-values ... \n");
            return 0;
        }
        if (index >= 0) {  // pvalues_for_family, cafe/pvalue.cpp:95-121
            prereqs(true, false);
            if (index >= fam.F()) throw std::runtime_error("ERROR(pvalue): family index out of range");
            std::vector<std::vector<double>> mats;
            const int S = fetch_matrices(mats);
            if (cond_dist.empty()) compute_conditional_distribution(mats, S);
            const int nl = tree.n_leaves(), ns = (int)fam.species.size();
            std::vector<int32_t> counts(nl, 0);
            for (int s_ = 0; s_ < ns; ++s_)
                if (species_index[s_] >= 0) counts[species_index[s_] / 2] = fam.counts[(size_t)index * ns + s_];
            const int32_t lo = range.root_min, hi = range.root_max, cm = range.max;
            std::vector<double> lh(hi - lo + 1);
            hip_check(cafehip_eval_root_likelihoods(ctx, 1, counts.data(), &lo, &hi, &cm, lh.data()));
            for (int i = 0; i <= hi - lo && shard_rank == 0; ++i)
                printf("%d\t%lg\t%lg\n", i + range.root_min, lh[i], pvalue_rank(lh[i], cond_dist[i].data(), num_random_samples));
            fflush(stdout);
            return 0;
        }
        throw std::runtime_error("pvalue: supported forms are -o file, -i file, -idx n");
    }

    int dispatch(const std::string& line_in)
    {
        std::string line = line_in;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        size_t b = 0;
        while (b < line.size() && isspace((unsigned char)line[b])) ++b;
        if (b >= line.size() || line[b] == '#') return 0;
        std::vector<std::string> tokens;
        {
            std::istringstream ss(line.substr(b));
            std::string t;
            while (ss >> t) tokens.push_back(t);
        }
        const std::string& cmd = tokens[0];
        if (cmd == "exit" || cmd == "quit") return 1;
        if (cmd == "seed") {  // cafe/cafe_commands.cpp:1950-1965
            if (tokens.size() < 2) throw std::runtime_error("Usage(seed): seed <value>");
            rng.seed((unsigned)atoi(tokens[1].c_str()));
            return 0;
        }
        if (cmd == "date") {  // :262-270
            time_t now = time(nullptr);
            log("%s", ctime(&now));
            return 0;
        }
        if (cmd == "echo") {
            std::string out;
            for (size_t i = 1; i < tokens.size(); ++i) out += (i > 1 ? " " : "") + tokens[i];
            log("%s\n", out.c_str());
            return 0;
        }
        if (cmd == "version") {
            log("Version: cafehip 0.1 (MI355X hot path for CAFE 4.2.1 semantics)\n");
            return 0;
        }
        if (cmd == "source") {
            if (tokens.size() < 2) throw std::runtime_error("Usage(source): source <file>");
            return run_script(tokens[1]);
        }
        if (cmd == "log") {  // cafe_cmd_log, cafe/cafe_commands.cpp:325-344
            if (tokens.size() == 1) {
                if (shard_rank == 0) printf("Log: %s\n", flog == stdout ? "stdout" : log_name.c_str());
                fflush(stdout);
                return 0;
            }
            std::string name;
            for (size_t i = 1; i < tokens.size(); ++i) name += (i > 1 ? " " : "") + tokens[i];
            if (name == "stdout") {
                if (own_log) fclose(flog);
                flog = stdout;
                own_log = false;
                log_name = "stdout";
            } else {
                FILE* f = fopen(name.c_str(), "a");
                if (!f) throw std::runtime_error("ERROR(log): Cannot open log file: " + name);
                if (own_log) fclose(flog);
                flog = f;
                own_log = true;
                log_name = name;
            }
            return 0;
        }
        if (cmd == "load") return cmd_load(tokens);
        if (cmd == "tree") return cmd_tree(tokens);
        if (cmd == "lambda") return cmd_lambda(tokens);
        if (cmd == "lambdamu") return cmd_lambdamu(tokens);
        if (cmd == "report") return cmd_report(tokens);
        if (cmd == "errormodel") return cmd_errormodel(tokens);
        if (cmd == "rootdist") return cmd_rootdist(tokens);
        if (cmd == "score") return cmd_score(tokens);
        if (cmd == "genfamily") return cmd_genfamily(tokens);
        if (cmd == "lhtest") return cmd_lhtest(tokens);
        if (cmd == "pvalue") return cmd_pvalue(tokens);
        if (cmd == "noerrormodel") {  // cafe/cafe_commands.cpp: remove the model from every species
            err_file.clear();
            err_mfs = -1;
            err_matrix.clear();
            err_leaf.clear();
            device_families_current = false;
            return 0;
        }
        throw std::runtime_error("command '" + cmd + "' is outside this build's scope (SURVEY.md section 8)");
    }

    int run_script(const std::string& path)
    {
        std::ifstream in(path);
        if (!in) throw std::runtime_error("Error(source): Cannot open " + path);
        std::string line;
        while (std::getline(in, line)) {
            const int rc = dispatch(line);
            if (rc != 0) return rc;
        }
        return 0;
    }
};

// ====================================================================================
// C API
// ====================================================================================
extern "C" {

const char* cafehost_last_error(void) { return g_host_err.c_str(); }

int cafehost_create(cafehost_session** out, int device_id, const char* log_path)
{
    if (!out) return host_fail("null out pointer");
    *out = nullptr;
    cafehost_session* s = new cafehost_session();
    s->device_id = device_id;
    if (cafehip_create(&s->ctx, device_id) != 0) {
        host_fail(std::string("cafehip: ") + cafehip_last_error());
        delete s;
        return -1;
    }
    if (log_path && strcmp(log_path, "stdout") != 0 && *log_path) {
        s->flog = fopen(log_path, "w");
        if (!s->flog) {
            cafehip_destroy(s->ctx);
            delete s;
            return host_fail(std::string("cannot open log file ") + log_path);
        }
        s->own_log = true;
        s->log_name = log_path;
    }
    // read once, here: nothing consults the environment during a command
    if (const char* e = getenv("CAFEHOST_SPECULATE")) s->opt_speculate = atoi(e) != 0;
    if (const char* e = getenv("CAFEHOST_LOOKAHEAD")) s->opt_lookahead = atoi(e) != 0;
    if (getenv("CAFEHOST_TIMING")) s->opt_timing = true;
    if (const char* e = getenv("CAFEHOST_LHTEST_DEAL")) s->opt_lhtest_deal = atoi(e);
    if (const char* e = getenv("CAFEHOST_GRID_DEAL")) s->opt_grid_deal = atoi(e);
    if (const char* e = getenv("CAFEHOST_PRIOR_LOOKAHEAD")) s->opt_prior_lookahead = atoi(e);
    if (const char* e = getenv("CAFEHOST_REPORT_ARITH")) s->opt_report_reference = std::string(e) == "reference";
    *out = s;
    return 0;
}

int cafehost_set_option(cafehost_session* s, const char* key, const char* value)
{
    if (!s || !key) return host_fail("null argument");
    const std::string k = key, v = value ? value : "";
    if (k == "speculate") {
        s->opt_speculate = (v.empty() || v == "auto") ? -1 : (atoi(v.c_str()) != 0);
        return 0;
    }
    if (k == "lookahead") {
        s->opt_lookahead = (v.empty() || v == "auto") ? -1 : (atoi(v.c_str()) != 0);
        return 0;
    }
    if (k == "timing") {
        s->opt_timing = atoi(v.c_str()) != 0;
        return 0;
    }
    if (k == "objective_arith") {
        if (v != "reference" && v != "fast" && !v.empty()) return host_fail("option objective_arith: fast | reference, got '" + v + "'");
        s->opt_objective_reference = v == "reference";
        s->spec.clear();
        return 0;
    }
    if (k == "report_arith") {
        if (v != "reference" && v != "fast" && !v.empty()) return host_fail("option report_arith: fast | reference, got '" + v + "'");
        s->opt_report_reference = v == "reference";
        return 0;
    }
    if (k == "prior_lookahead") {
        s->opt_prior_lookahead = atoi(v.c_str());
        return 0;
    }
    if (k == "lhtest_deal") {
        s->opt_lhtest_deal = atoi(v.c_str());
        return 0;
    }
    if (k == "grid_deal") {
        s->opt_grid_deal = atoi(v.c_str());
        return 0;
    }
    if (k == "prior_file") {
        s->opt_prior_file = v;
        s->spec.clear();
        return 0;
    }
    // everything else is a switch of the device context(s)
    if (cafehip_set_option(s->ctx, key, v.c_str()) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    if (s->ctx_one && cafehip_set_option(s->ctx_one, key, v.c_str()) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    s->spec.clear();
    return 0;
}

int cafehost_rng_selftest(unsigned seed, int n_before, int n_bulk, int n_after)
{
    // the bulk draw of the Monte-Carlo null against glibc's own random_r, single draws on either side of it
    GlibcRand a, b;
    a.seed(seed);
    b.seed(seed);
    const int n = std::max(n_before, 0) + std::max(n_bulk, 0) + std::max(n_after, 0);
    std::vector<int32_t> want(n), got(n);
    for (int i = 0; i < n; ++i) random_r(&a.rd, &want[i]);
    int k = 0;
    for (int i = 0; i < n_before; ++i) random_r(&b.rd, &got[k++]);
    if (n_bulk > 0) b.fill_raw(got.data() + k, (size_t)n_bulk);
    k += std::max(n_bulk, 0);
    for (int i = 0; i < n_after; ++i) random_r(&b.rd, &got[k++]);
    for (int i = 0; i < n; ++i)
        if (want[i] != got[i]) return host_fail("bulk random stream differs from random_r at draw " + std::to_string(i));
    return 0;
}

int cafehost_poisson_fit_selftest(const int* leaf_sizes, long n, double start, int lookahead, double* lambda, double* score, int* iters,
                                  long* passes)
{
    if (!leaf_sizes || n < 0 || !lambda || !score) return host_fail("bad argument");
    PoissonFit fit;
    fit.run(std::vector<int>(leaf_sizes, leaf_sizes + n), start, lookahead != 0);
    *lambda = fit.lambda;
    *score = fit.score;
    if (iters) *iters = fit.iters;
    if (passes) *passes = fit.passes + fit.misses;   // sweeps over the table
    return 0;
}

double cafehost_pvalue_selftest(double v, const double* sorted_null, int size) { return pvalue_rank(v, sorted_null, size); }

// the report's number formatting against printf: returns how many of `n` values differ (0 expected), the first in `first_bad`
long cafehost_format_selftest(const double* values, long n, double* first_bad)
{
    long bad = 0;
    std::string fast;
    char slow[64];
    for (long i = 0; i < n; ++i) {
        fast.clear();
        append_g(fast, values[i]);
        snprintf(slow, sizeof slow, "%g", values[i]);
        bool same = fast == slow;
        const int iv = (int)std::max(-2147483648.0, std::min(2147483647.0, values[i] * 1e6));
        fast.clear();
        append_int(fast, iv);
        snprintf(slow, sizeof slow, "%d", iv);
        same = same && fast == slow;
        if (!same) {
            if (bad == 0 && first_bad) *first_bad = values[i];
            ++bad;
        }
    }
    return bad;
}

int cafehost_fminsearch_selftest(cafehost_math_fn eq, int n, void* args, const double* x0, double tolx, double tolf, double* xmin,
                                 double* fmin, int* bymax)
{
    if (!eq || !x0 || !xmin || !fmin || n < 1) return host_fail("bad argument");
    FMinSearch pfm;
    pfm.init(n);
    pfm.tolx = tolx;
    pfm.tolf = tolf;
    std::vector<double> x(n);
    pfm.eq = [&](const double* p) {
        std::copy(p, p + n, x.begin());
        return eq(x.data(), args);
    };
    pfm.minimize(x0);
    std::copy(pfm.v[0].begin(), pfm.v[0].end(), xmin);
    *fmin = pfm.fv[0];
    if (bymax) *bymax = pfm.bymax ? 1 : 0;
    return pfm.iters;
}

// Test hook (host only): the same minimisation with the look-ahead hook installed.  out = {evaluations, evaluations whose
// point was among the twelve most recently announced before the PREVIOUS evaluations (bit for bit), announcements, points}.
int cafehost_lookahead_selftest(cafehost_math_fn eq, int n, void* args, const double* x0, double tolx, double tolf, double* xmin,
                                double* fmin, long out[4])
{
    if (!eq || !x0 || !xmin || !fmin || !out || n < 1) return host_fail("bad argument");
    FMinSearch pfm;
    pfm.init(n);
    pfm.tolx = tolx;
    pfm.tolf = tolf;
    std::vector<double> x(n);
    // what the device store would hold: the points of the announcements made before earlier evaluations, the twelve most
    // recent ones (an announcement made before THIS evaluation is built beside it and serves the next)
    std::vector<std::vector<double>> pending, store;
    long evals = 0, covered = 0, calls = 0, points = 0;
    pfm.lookahead = [&](const std::vector<std::vector<double>>& pts) {
        pending = pts;
        ++calls;
        points += (long)pts.size();
    };
    pfm.eq = [&](const double* p) {
        std::copy(p, p + n, x.begin());
        for (const auto& q : store)
            if (std::equal(q.begin(), q.end(), p)) {
                ++covered;
                break;
            }
        ++evals;
        for (const auto& q : pending) {
            bool have = false;
            for (const auto& r : store) have = have || r == q;
            if (!have) store.push_back(q);
        }
        pending.clear();
        while (store.size() > 12) store.erase(store.begin());
        return eq(x.data(), args);
    };
    pfm.minimize(x0);
    std::copy(pfm.v[0].begin(), pfm.v[0].end(), xmin);
    *fmin = pfm.fv[0];
    out[0] = evals;
    out[1] = covered;
    out[2] = calls;
    out[3] = points;
    return pfm.iters;
}

void cafehost_destroy(cafehost_session* s)
{
    if (!s) return;
    s->drop_prior_job();
    if (s->own_log && s->flog) fclose(s->flog);
    cafehip_destroy(s->ctx);
    if (s->ctx_one) cafehip_destroy(s->ctx_one);
    delete s;
}

int cafehost_dispatch(cafehost_session* s, const char* command_line)
{
    if (!s || !command_line) return host_fail("null argument");
    try {
        const int rc = s->dispatch(command_line);
        s->drop_prior_job();   // (normally picked up by the command itself)
        return rc;
    } catch (const std::exception& e) {  // cafe/cafe_commands.cpp:531-535
        s->drop_prior_job();
        return host_fail(e.what());
    }
}

int cafehost_run_script(cafehost_session* s, const char* path)
{
    if (!s || !path) return host_fail("null argument");
    try {
        const int rc = s->run_script(path);
        s->drop_prior_job();
        return rc;
    } catch (const std::exception& e) {
        s->drop_prior_job();
        return host_fail(e.what());
    }
}

int cafehost_set_shard(cafehost_session* s, int rank, int world)
{
    if (!s) return host_fail("null session");
    if (world < 1 || rank < 0 || rank >= world) return host_fail("bad shard " + std::to_string(rank) + "/" + std::to_string(world));
    s->shard_rank = rank;
    s->shard_world = world;
    s->device_families_current = false;
    return 0;
}

int cafehost_shard_bounds(cafehost_session* s, int* lo, int* hi, int* n_chunks_local)
{
    if (!s) return host_fail("null session");
    if (lo) *lo = s->shard_lo;
    if (hi) *hi = s->shard_hi;
    if (n_chunks_local) *n_chunks_local = (s->shard_hi - s->shard_lo + CAFEHIP_CHUNK - 1) / CAFEHIP_CHUNK;
    return 0;
}

int cafehost_set_exchange(cafehost_session* s, cafehost_exchange_fn exchange, void* user, void* d_chunk_sums,
                          void* d_first_zero)
{
    if (!s) return host_fail("null session");
    if (exchange && (!d_chunk_sums || !d_first_zero)) return host_fail("exchange needs device buffers");
    s->exchange = exchange;
    s->exchange_user = user;
    s->d_exch_chunks = (double*)d_chunk_sums;
    s->d_exch_fz = (int32_t*)d_first_zero;
    return 0;
}

int cafehost_set_allgather(cafehost_session* s, cafehost_allgather_fn fn, void* user)
{
    if (!s) return host_fail("null session");
    s->allgather = fn;
    s->allgather_user = user;
    return 0;
}

int cafehost_comm_unique_id(void* out_id)
{
    if (!out_id) return host_fail("null id buffer");
    static_assert(CAFEHOST_COMM_ID_BYTES == CAFEHIP_COMM_ID_BYTES, "one id size");
    if (cafehip_comm_unique_id(out_id) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    return 0;
}

int cafehost_comm_cleanup(const void* unique_id)
{
    if (!unique_id) return host_fail("null communicator id");
    return cafehip_comm_cleanup(unique_id);
}

int cafehost_init_comm(cafehost_session* s, int rank, int world, const void* unique_id)
{
    if (!s) return host_fail("null session");
    if (!unique_id) return host_fail("null communicator id");
    if (world < 1 || rank < 0 || rank >= world) return host_fail("bad rank " + std::to_string(rank) + "/" + std::to_string(world));
    if (s->native_comm) return host_fail("this session already has a communicator");
    if (cafehip_comm_init(s->ctx, rank, world, unique_id) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    s->native_comm = true;
    s->allgather = &cafehost_session::native_allgather;   // (host blocks: report phase, genfamily's barrier, lhtest's lines)
    s->allgather_user = s;
    s->shard_rank = rank;
    s->shard_world = world;
    s->device_families_current = false;   // the next upload shards the table and wires the exchange
    return 0;
}

int cafehost_speculation_stats(cafehost_session* s, long* launches, long* points, long* hits)
{
    if (!s) return host_fail("null session");
    if (launches) *launches = s->spec_launches;
    if (points) *points = s->spec_points;
    if (hits) *hits = s->spec_hits;
    return 0;
}

int cafehost_lookahead_stats(cafehost_session* s, long out[4])
{
    if (!s || !out) return host_fail("null argument");
    long mc[CAFEHIP_MATRIX_CACHE_STATS] = {};
    if (cafehip_matrix_cache_stats(s->ctx, mc) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    out[0] = s->look_calls;
    out[1] = s->look_points;
    out[2] = mc[2];   // evaluations that found their matrices on the device
    out[3] = mc[1];   // parameter sets whose matrices were built ahead of time
    return 0;
}

int cafehost_exchange_stats(cafehost_session* s, double* seconds, long* calls)
{
    if (!s) return host_fail("null session");
    double sec = 0;
    long n = 0;
    if (cafehip_comm_info(s->ctx, nullptr, nullptr, nullptr, nullptr, &sec, &n) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    if (seconds) *seconds = sec;
    if (calls) *calls = n;
    return 0;
}

int cafehost_fetch_small(cafehost_session* s, const void* d_src, unsigned long nbytes, const void** host_ptr)
{
    if (!s) return host_fail("null session");
    if (cafehip_fetch_small(s->ctx, d_src, (size_t)nbytes, host_ptr) != 0)
        return host_fail(std::string("cafehip: ") + cafehip_last_error());
    return 0;
}

int cafehost_set_stream(cafehost_session* s, void* hip_stream)
{
    if (!s) return host_fail("null session");
    if (cafehip_set_stream(s->ctx, hip_stream) != 0) return host_fail(std::string("cafehip: ") + cafehip_last_error());
    return 0;
}

int cafehost_upload(cafehost_session* s)
{
    if (!s) return host_fail("null session");
    try {
        s->prereqs(true, true);
        s->upload();
        return 0;
    } catch (const std::exception& e) {
        return host_fail(e.what());
    }
}

int cafehost_num_params(cafehost_session* s) { return s ? s->num_params : -1; }

int cafehost_get_params(cafehost_session* s, double* out, int n)
{
    if (!s || !out) return host_fail("null argument");
    const int m = std::min<int>(n, (int)s->params.size());
    for (int i = 0; i < m; ++i) out[i] = s->params[i];
    return m;
}

double cafehost_last_score(cafehost_session* s) { return s ? s->last_score : NAN; }
int cafehost_search_iterations(cafehost_session* s) { return s ? s->search_iters : -1; }
int cafehost_num_evaluations(cafehost_session* s) { return s ? s->n_evals : -1; }
double cafehost_search_seconds(cafehost_session* s) { return s ? s->search_seconds : NAN; }
double cafehost_poisson_lambda(cafehost_session* s) { return s ? s->poisson_lambda : NAN; }

int cafehost_get_trace(cafehost_session* s, double* out, int max_rows)
{
    if (!s || !out) return host_fail("null argument");
    const int w = s->num_params + 1;
    const int rows = std::min<int>(max_rows, w ? (int)s->trace.size() / w : 0);
    std::copy(s->trace.begin(), s->trace.begin() + (size_t)rows * w, out);
    return rows;
}

}  // extern "C"
