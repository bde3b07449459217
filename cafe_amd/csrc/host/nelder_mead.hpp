// nelder_mead.hpp -- Nelder-Mead exactly as libcommon/fminsearch.cpp, with the hook that tells the owner which points may be asked for next
// (part of the host driver, cafe_host.cpp; split out in round 4 so that the session file holds the commands only)
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

namespace cafehost_impl {

// ------------------------------------------------------------------------------------
// Nelder-Mead exactly as libcommon/fminsearch.cpp (defaults :7-21, loop :264-302)
// ------------------------------------------------------------------------------------
struct FMinSearch {
    int N = 0, N1 = 0, maxiters = 10000, iters = 0, bymax = 0;
    double rho = 1, chi = 2, psi = 0.5, sigma = 0.5, tolx = 1e-6, tolf = 1e-6, delta = 0.05, zero_delta = 0.00025;
    std::vector<std::vector<double>> v, vsort;
    std::vector<double> fv, x_mean, x_r, x_tmp;
    std::vector<int> idx;
    std::function<double(const double*)> eq;
    // Optional: told which points the NEXT calls of eq() may ask for -- the four candidates of an iteration
    // (reflection, expansion, the two contractions: all functions of the current simplex,
    // libcommon/fminsearch.cpp:198-237), the initial simplex, the vertices of a shrink -- so that the owner can
    // evaluate them in one batched device pass.  eq() is still called in the reference's order with the reference's
    // accept rules; the hook only changes where the values come from.
    std::function<void(const std::vector<std::vector<double>>&)> prefetch;
    // Optional (round 5): told, just BEFORE a call of eq(), which points the call AFTER it may ask for, whatever value the
    // pending one returns -- the expansion / contraction point of the reflection being evaluated, the reflection of every
    // simplex the pending decision can leave behind, the vertices of a shrink.  They are built with the statements of
    // minimize() itself (candidates(), shrink()), so they are bit-identical to what is asked for later.  The owner can
    // have the transition matrices of those points built while eq() runs (cafehip_prefetch_matrices): the optimiser's
    // decisions, its call order and every value stay those of the plain loop.
    std::function<void(const std::vector<std::vector<double>>&)> lookahead;

    void init(int n)
    {
        N = n;
        N1 = n + 1;
        v.assign(N1, std::vector<double>(N, 0.0));
        vsort = v;
        fv.assign(N1, 0.0);
        x_mean.assign(N, 0.0);
        x_r.assign(N, 0.0);
        x_tmp.assign(N, 0.0);
        idx.assign(N1, 0);
    }

    // Order fv ascending, idx riding along (libcommon/fminsearch.cpp:77-107 does this job).  NOT a stable sort, and which
    // of two equal values ends up first steers the simplex (the worst vertex is the one replaced), so the permutation has
    // to be the reference's: a hole-moving partition around the FIRST element of a span, the hole alternating between
    // the low and the high end.  Restated here with an explicit work list; comparisons are written as the reference has
    // them (`key <= x`, `key >= x`: a NaN score -- the k-cluster objective can return one -- must stop both scans).
    void order_values_with_index()
    {
        struct Span { int lo, hi; };
        std::vector<Span> work;
        work.push_back(Span{0, N});
        while (!work.empty()) {
            const Span span = work.back();
            work.pop_back();
            if (span.lo >= span.hi) continue;
            const double key = fv[span.lo];
            const int key_id = idx[span.lo];
            int a = span.lo, b = span.hi;   // unsettled part; the hole is at a (low phase) or at b (high phase)
            for (;;) {
                while (a < b && key <= fv[b]) --b;     // from the top: first value below the key ...
                if (a == b) break;
                fv[a] = fv[b];                         // ... drops into the hole at the bottom; the hole is at b now
                idx[a] = idx[b];
                ++a;
                while (a < b && key >= fv[a]) ++a;     // from the bottom: first value above the key ...
                if (a == b) break;
                fv[b] = fv[a];                         // ... rises into the hole at the top; the hole is at a again
                idx[b] = idx[a];
                --b;
            }
            fv[a] = key;
            idx[a] = key_id;
            work.push_back(Span{span.lo, a - 1});      // the two sides are disjoint: their order does not matter
            work.push_back(Span{a + 1, span.hi});
        }
    }

    void sort()
    {  // __fminsearch_sort :109-123
        for (int i = 0; i < N1; ++i) idx[i] = i;
        order_values_with_index();
        for (int i = 0; i < N1; ++i) vsort[i] = v[idx[i]];
        v = vsort;
    }

    bool checkV() const
    {  // :126-141
        double mx = -1.7976931348623157e+308;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) mx = std::max(mx, std::fabs(v[i + 1][j] - v[i][j]));
        return mx <= tolx;
    }

    bool checkF() const
    {  // :143-154
        double mx = -1.7976931348623157e+308;
        for (int i = 1; i < N1; ++i) mx = std::max(mx, std::fabs(fv[i] - fv[0]));
        return mx <= tolf;
    }

    void set_last(const std::vector<double>& x, double f)
    {  // :252-262
        v[N] = x;
        fv[N] = f;
        sort();
    }

    void shrink()
    {  // :238-250
        for (int i = 1; i < N1; ++i)
            for (int j = 0; j < N; ++j) v[i][j] = v[0][j] + sigma * (v[i][j] - v[0][j]);
        if (prefetch) prefetch(std::vector<std::vector<double>>(v.begin() + 1, v.end()));
        for (int i = 1; i < N1; ++i) {
            if (lookahead && i == N) announce_after_last_vertex();
            fv[i] = eq(v[i].data());
        }
        sort();
    }

    // ---- look-ahead (see `lookahead` above) ----------------------------------------------------------------------
    // the sorted simplex with its worst vertex dropped and p at position k (0 = new best ... N = new worst)
    std::vector<std::vector<double>> with_vertex(const std::vector<double>& p, int k) const
    {
        std::vector<std::vector<double>> sv;
        sv.reserve(N1);
        for (int j = 0; j < N; ++j) {
            if (j == k) sv.push_back(p);
            sv.push_back(v[j]);
        }
        if (k >= N) sv.push_back(p);
        return sv;
    }
    // reflections of the simplexes that accepting p can produce: p lands at position k when its value falls between the
    // k-th and the (k+1)-th of the vertices that stay (a position between two EQUAL values needs a tie: not announced)
    void reflections_if_accepted(const std::vector<double>& p, int k_first, std::vector<std::vector<double>>& out) const
    {
        for (int k = k_first; k <= N; ++k) {
            if (k > 0 && !(fv[k - 1] < fv[k])) continue;
            out.push_back(candidates(with_vertex(p, k))[0]);
        }
    }
    void shrink_vertices(std::vector<std::vector<double>>& out) const
    {
        for (int i = 1; i < N1; ++i) {
            std::vector<double> q(N);
            for (int j = 0; j < N; ++j) q[j] = v[0][j] + sigma * (v[i][j] - v[0][j]);
            out.push_back(q);
        }
    }
    // before a contraction point is evaluated: it is accepted somewhere in the order, or the simplex shrinks
    void announce_after_contraction(const std::vector<double>& p)
    {
        std::vector<std::vector<double>> pts;
        reflections_if_accepted(p, 0, pts);
        shrink_vertices(pts);
        lookahead(pts);
    }
    // before the LAST vertex of the initial simplex or of a shrink is evaluated: what follows is the reflection of the
    // sorted simplex, whose order depends on the value to come.  With one parameter there are two orders; with more the
    // orders multiply and nothing is announced.
    void announce_after_last_vertex()
    {
        if (N != 1) return;
        std::vector<std::vector<double>> pts;
        pts.push_back(candidates({v[0], v[1]})[0]);
        pts.push_back(candidates({v[1], v[0]})[0]);
        lookahead(pts);
    }

    // The four points an iteration may evaluate for the SORTED simplex `sv` (reflection, expansion, inside and outside
    // contraction: libcommon/fminsearch.cpp:189-237) -- the arithmetic of minimize() below, so that a caller looking
    // ahead asks for bit-identical points.
    std::vector<std::vector<double>> candidates(const std::vector<std::vector<double>>& sv) const
    {
        std::vector<std::vector<double>> pts(4, std::vector<double>(N));
        for (int a = 0; a < N; ++a) {
            double mean = 0;
            for (int j = 0; j < N; ++j) mean += sv[j][a];
            mean /= N;
            const double xr = mean + rho * (mean - sv[N][a]);
            pts[0][a] = xr;
            pts[1][a] = mean + chi * (xr - mean);
            pts[2][a] = mean + psi * (mean - sv[N][a]);
            pts[3][a] = mean + psi * (xr - mean);
        }
        return pts;
    }

    int minimize(const double* X0)
    {
        // __fminsearch_min_init :156-187 (note the isinf(previous vertex) rule)
        if (prefetch) {
            // the simplex as it comes out when no vertex evaluates to infinity (otherwise some points differ and are
            // simply evaluated on demand)
            std::vector<std::vector<double>> pts(N1, std::vector<double>(N));
            for (int i = 0; i < N1; ++i)
                for (int j = 0; j < N; ++j) pts[i][j] = ((i - 1) == j) ? (X0[j] ? (1 + delta) * X0[j] : zero_delta) : X0[j];
            prefetch(pts);
        }
        for (int i = 0; i < N1; ++i) {
            for (int j = 0; j < N; ++j) {
                const bool big = (i > 1 && std::isinf(fv[i - 1]));
                if ((i - 1) == j)
                    v[i][j] = X0[j] ? (1 + (big ? delta * 100 : delta)) * X0[j] : zero_delta;
                else
                    v[i][j] = X0[j];
            }
            if (lookahead && i == 0) {
                // the other vertices of the initial simplex (as they come out when no vertex evaluates to infinity)
                std::vector<std::vector<double>> pts(N, std::vector<double>(N));
                for (int q = 1; q < N1; ++q)
                    for (int j = 0; j < N; ++j) pts[q - 1][j] = ((q - 1) == j) ? (X0[j] ? (1 + delta) * X0[j] : zero_delta) : X0[j];
                lookahead(pts);
            } else if (lookahead && i == N) {
                announce_after_last_vertex();
            }
            fv[i] = eq(v[i].data());
        }
        sort();
        int i;
        for (i = 0; i < maxiters; ++i) {
            if (checkV() && checkF()) break;
            for (int a = 0; a < N; ++a) {  // x_mean :189-201
                x_mean[a] = 0;
                for (int j = 0; j < N; ++j) x_mean[a] += v[j][a];
                x_mean[a] /= N;
            }
            for (int a = 0; a < N; ++a) x_r[a] = x_mean[a] + rho * (x_mean[a] - v[N][a]);
            if (prefetch) {
                std::vector<std::vector<double>> pts(4, std::vector<double>(N));
                for (int a = 0; a < N; ++a) {
                    pts[0][a] = x_r[a];
                    pts[1][a] = x_mean[a] + chi * (x_r[a] - x_mean[a]);     // expansion
                    pts[2][a] = x_mean[a] + psi * (x_mean[a] - v[N][a]);   // inside contraction
                    pts[3][a] = x_mean[a] + psi * (x_r[a] - x_mean[a]);    // outside contraction
                }
                prefetch(pts);
            }
            // the reference's accept rules (libcommon/fminsearch.cpp:203-237), one decision per outcome of the reflection:
            // better than the best -> try the expansion; no better than the worst -> contract (inside when strictly
            // worse, outside on a tie) or shrink; anything in between -> take the reflection
            if (lookahead) {
                // after the reflection: the expansion (better than the best), the inside contraction (worse than the worst),
                // or -- the reflection accepted -- the next iteration's reflection (the outside contraction needs a tie)
                std::vector<std::vector<double>> pts(2, std::vector<double>(N));
                for (int a = 0; a < N; ++a) {
                    pts[0][a] = x_mean[a] + chi * (x_r[a] - x_mean[a]);
                    pts[1][a] = x_mean[a] + psi * (x_mean[a] - v[N][a]);
                }
                reflections_if_accepted(x_r, 1, pts);
                lookahead(pts);
            }
            const double f_reflect = eq(x_r.data());
            const double f_best = fv[0], f_worst = fv[N];
            if (f_reflect < f_best) {
                for (int a = 0; a < N; ++a) x_tmp[a] = x_mean[a] + chi * (x_r[a] - x_mean[a]);
                if (lookahead) {
                    // whichever of the two is kept becomes the new best vertex
                    std::vector<std::vector<double>> pts;
                    pts.push_back(candidates(with_vertex(x_tmp, 0))[0]);
                    pts.push_back(candidates(with_vertex(x_r, 0))[0]);
                    lookahead(pts);
                }
                const double f_expand = eq(x_tmp.data());
                if (f_expand < f_reflect) set_last(x_tmp, f_expand);
                else set_last(x_r, f_reflect);
            } else if (f_reflect > f_worst) {
                for (int a = 0; a < N; ++a) x_tmp[a] = x_mean[a] + psi * (x_mean[a] - v[N][a]);
                if (lookahead) announce_after_contraction(x_tmp);
                const double f_inside = eq(x_tmp.data());
                if (f_inside < f_worst) set_last(x_tmp, f_inside);
                else shrink();
            } else if (f_reflect >= f_worst) {   // == the worst (a NaN fails both tests above and this one: next branch)
                for (int a = 0; a < N; ++a) x_tmp[a] = x_mean[a] + psi * (x_r[a] - x_mean[a]);
                if (lookahead) announce_after_contraction(x_tmp);
                const double f_outside = eq(x_tmp.data());
                if (f_outside <= f_reflect) set_last(x_tmp, f_outside);
                else shrink();
            } else {
                set_last(x_r, f_reflect);
            }
        }
        bymax = (i == maxiters);
        iters = i;
        return bymax;
    }
};

}  // namespace cafehost_impl
