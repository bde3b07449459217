// poisson_prior.hpp -- the empirical root-size prior: poisspdf, the Poisson fit's objective as chains of additions, the fit with look-ahead
// (part of the host driver, cafe_host.cpp; split out in round 4 so that the session file holds the commands only)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <utility>
#include <vector>

#include "../host_math.hpp"
#include "nelder_mead.hpp"

namespace cafehost_impl {

// poisspdf, libcommon/mathfunc.c:352-355
inline double poisspdf(int x, double lambda) { return std::exp(x * std::log(lambda) - cafehip::gammaln(x + 1) - lambda); }

// ------------------------------------------------------------------------------------
// Objective of the empirical root-size prior (__lnLPoisson, cafe/lambda.cpp:771-787): -sum_i log poisspdf(x_i, lambda)
// over every non-zero leaf count of the table, added in table order.  The sum is one long chain of dependent additions
// (4 million of them for the configs[3] shard: ~3 ms per call, ~70 calls per fit -- more than the whole GPU search at
// that size), and its rounding sequence is what the reference computes, so it is kept addition for addition.  What
// CAN be done is to evaluate SEVERAL lambdas per pass: the chains are independent, so a core interleaves a few at the
// latency of one, and the cores take different ones.  Which lambdas: the 1-D Nelder-Mead only ever asks for points
// that are fixed functions of its two vertices -- the four candidates of the current iteration and, for each of the
// nine ordered simplices the iteration can end in, the four candidates of the next one.  prefetch() evaluates those in
// one pass (every second iteration then finds all it needs in the cache); value() answers from the cache or, on a miss,
// with a single chain.  Same values, same call order, same fitted lambda: only where the values come from changes.
// ------------------------------------------------------------------------------------
struct PoissonChains {
    std::vector<uint16_t> sizes16;   // x_i (count - 1), table order
    std::vector<int> sizes32;        // ... when one does not fit 16 bits
    int max_x = 0;
    std::vector<std::pair<double, double>> cache;   // (lambda, -score)
    long passes = 0, chains = 0, hits = 0, misses = 0;

    void set(const std::vector<int>& leaf_sizes)
    {
        max_x = 0;
        for (int x : leaf_sizes) max_x = std::max(max_x, x);
        sizes16.clear();
        sizes32.clear();
        if (max_x < 65536) sizes16.assign(leaf_sizes.begin(), leaf_sizes.end());
        else sizes32 = leaf_sizes;
        cache.clear();
    }
    size_t n() const { return sizes32.empty() ? sizes16.size() : sizes32.size(); }

    // log(poisspdf(x, lambda)) is a pure function of x: once per distinct size, as the reference's loop body has it
    void terms_of(double lambda, double* term) const
    {
        for (int x = 0; x <= max_x; ++x) {
            double ll = poisspdf(x, lambda);
            if (std::isnan(ll)) ll = 0;
            term[x] = std::log(ll);
        }
    }

    // Up to 8 chains in one sweep: eight scalar accumulators, each receiving ITS chain's additions in table order; the
    // eight dependency chains overlap in the core's pipeline, so the sweep costs what one chain costs (measured: 3.0 ms
    // for 4 M sizes with 1 or with 8 chains).  Unused lanes point at lane 0's table and are ignored.
    template <class T>
    static void sweep8(const T* xs, size_t n, const double* const* term, double* score)
    {
        const double *t0 = term[0], *t1 = term[1], *t2 = term[2], *t3 = term[3], *t4 = term[4], *t5 = term[5], *t6 = term[6], *t7 = term[7];
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
        for (size_t i = 0; i < n; ++i) {
            const size_t x = xs[i];
            s0 += t0[x];
            s1 += t1[x];
            s2 += t2[x];
            s3 += t3[x];
            s4 += t4[x];
            s5 += t5[x];
            s6 += t6[x];
            s7 += t7[x];
        }
        score[0] = s0, score[1] = s1, score[2] = s2, score[3] = s3, score[4] = s4, score[5] = s5, score[6] = s6, score[7] = s7;
    }
    template <class T>
    static void sweep1(const T* xs, size_t n, const double* term, double* score)
    {
        double s = 0;
        for (size_t i = 0; i < n; ++i) s += term[xs[i]];
        *score = s;
    }

    static constexpr int kLanes = 8;
    template <class T>
    void run_group(const T* xs, const double* lambdas, int count, double* out) const
    {
        std::vector<double> tabs((size_t)count * (max_x + 1));
        const double* term[kLanes];
        double score[kLanes];
        for (int k = 0; k < kLanes; ++k) {
            if (k < count) terms_of(lambdas[k], tabs.data() + (size_t)k * (max_x + 1));
            term[k] = tabs.data() + (size_t)(k < count ? k : 0) * (max_x + 1);
        }
        if (count == 1) sweep1(xs, n(), term[0], score);
        else sweep8(xs, n(), term, score);
        for (int k = 0; k < count; ++k) out[k] = -score[k];
    }

    void evaluate(const double* lambdas, int count, double* out) const
    {
        for (int k0 = 0; k0 < count; k0 += kLanes) {
            const int g = std::min(kLanes, count - k0);
            if (sizes32.empty()) run_group(sizes16.data(), lambdas + k0, g, out + k0);
            else run_group(sizes32.data(), lambdas + k0, g, out + k0);
        }
    }

    bool lookup(double lambda, double* f) const
    {
        for (auto it = cache.rbegin(); it != cache.rend(); ++it)
            if (it->first == lambda) {
                *f = it->second;
                return true;
            }
        return false;
    }

    double value(double lambda)
    {
        double f;
        if (lookup(lambda, &f)) {
            ++hits;
            return f;
        }
        ++misses;
        evaluate(&lambda, 1, &f);
        cache.emplace_back(lambda, f);
        return f;
    }

    // evaluate every point not yet known, the chains dealt evenly to up to 16 cores
    void prefetch(std::vector<double> want)
    {
        std::vector<double> todo;
        for (double x : want) {
            double f;
            if (std::isnan(x) || lookup(x, &f) || std::find(todo.begin(), todo.end(), x) != todo.end()) continue;
            todo.push_back(x);
        }
        if (todo.empty()) return;
        if (cache.size() > 512) cache.erase(cache.begin(), cache.begin() + 256);
        std::vector<double> f(todo.size());
        const int hw = std::max(1u, std::thread::hardware_concurrency());
        // small tables: one core runs everything (a thread costs more than their chains)
        // (measured on the MI355X box's host, 4 M sizes: a one-chain sweep 1.7 ms, an eight-chain sweep 3.9 ms -- a chain per
        // core while cores last, several per core only beyond that)
        const int workers = (n() < 200000) ? 1 : std::max(1, std::min<int>({hw, 16, (int)todo.size()}));
        if (workers == 1) {
            evaluate(todo.data(), (int)todo.size(), f.data());
        } else {
            std::vector<std::thread> pool;
            for (int w = 0; w < workers; ++w) {
                const int a = (int)((long long)todo.size() * w / workers), b = (int)((long long)todo.size() * (w + 1) / workers);
                if (b > a) pool.emplace_back([&, a, b] { evaluate(todo.data() + a, b - a, f.data() + a); });
            }
            for (auto& t : pool) t.join();
        }
        for (size_t i = 0; i < todo.size(); ++i) cache.emplace_back(todo[i], f[i]);
        ++passes;
        chains += (long)todo.size();
    }
};


// find_poisson_lambda (cafe/lambda.cpp:808-838): 1-D Nelder-Mead on the objective above from the given start
struct PoissonFit {
    static constexpr size_t kLookaheadMinSizes = 500000;
    double lambda = 0, score = 0;
    int iters = 0;
    long passes = 0, chains = 0, hits = 0, misses = 0;

    void run(const std::vector<int>& leaf_sizes, double start, bool lookahead)
    {
        FMinSearch pfm;
        pfm.init(1);
        pfm.tolx = 1e-6;
        pfm.tolf = 1e-6;
        PoissonChains ch;
        ch.set(leaf_sizes);
        pfm.eq = [&](const double* pl) { return ch.value(pl[0]); };   // __lnLPoisson :771-787
        bool looked_ahead = false;
        // (below ~half a million sizes a sweep is tens of microseconds: one chain per call is as fast as it gets --
        // measured 4.1 ms plain against 5.4 ms with look-ahead at 160 k sizes, 46 against 25 ms at 1.2 M)
        if (lookahead && leaf_sizes.size() >= kLookaheadMinSizes)
            pfm.prefetch = [&](const std::vector<std::vector<double>>& pts) {
                std::vector<double> want;
                for (auto& p_ : pts) want.push_back(p_[0]);
                if (pts.size() == 4) {
                    // the four candidates of this iteration; every second time also those of the NEXT one, for each ordered
                    // simplex this iteration can end in: the best vertex a with one of {reflection, expansion, the two
                    // contractions, the shrunk worst vertex} on either side of it (FMinSearch::candidates: same arithmetic)
                    bool known = true;
                    double f_;
                    for (double x : want) known = known && ch.lookup(x, &f_);
                    if (!(known && looked_ahead)) {
                        const double a = pfm.v[0][0], b = pfm.v[1][0];
                        const double shrunk = a + pfm.sigma * (b - a);
                        want.push_back(shrunk);
                        const double fresh[5] = {pts[0][0], pts[1][0], pts[2][0], pts[3][0], shrunk};
                        for (double x : fresh)
                            for (int order = 0; order < 2; ++order) {
                                const std::vector<std::vector<double>> simplex =
                                    order ? std::vector<std::vector<double>>{{a}, {x}} : std::vector<std::vector<double>>{{x}, {a}};
                                for (auto& c_ : pfm.candidates(simplex)) want.push_back(c_[0]);
                            }
                        looked_ahead = true;
                    } else {
                        looked_ahead = false;   // everything this iteration needs came from the last look-ahead: no pass
                    }
                }
                ch.prefetch(want);
            };
        pfm.minimize(&start);
        lambda = pfm.v[0][0];
        score = pfm.fv[0];
        iters = pfm.iters;
        passes = ch.passes;
        chains = ch.chains;
        hits = ch.hits;
        misses = ch.misses;
    }
};

}  // namespace cafehost_impl
