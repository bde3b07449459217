// host_util.hpp -- number formatting as the reference prints it, the rank of a likelihood in a sorted null, command-line argument lists
// (part of the host driver, cafe_host.cpp; split out in round 4 so that the session file holds the commands only)
#pragma once
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace cafehost_impl {

inline std::string join_double(const double* v, int n)
{  // string_pchar_join_double, libcommon/utils_string.c:227-237
    std::string out;
    char buf[64];
    for (int i = 0; i < n; ++i) {
        snprintf(buf, sizeof buf, "%15.14lf", v[i]);
        out += buf;
        if (i < n - 1) out += ",";
    }
    return out;
}

// pvalue(), libcommon/mathfunc.c:663-689: rank of v in the ascending null sample, ties split in half
inline double pvalue_rank(double v, const double* conddist, int size)
{
    // conddist is sorted ascending (the caller sorts the null's likelihoods).  With `below` values smaller than v and a
    // run of `equal` values equal to it, the reference's search ends on (first, last) of that run and returns
    // (first + 1 + (last - first) / 2) / size -- the middle of the run, one-based -- or below / size when nothing equals v.
    if (size <= 0) return 0.0;
    const double* const end = conddist + size;
    const double* const first_not_below = std::lower_bound(conddist, end, v);
    const double* const first_above = std::upper_bound(first_not_below, end, v);
    const int below = (int)(first_not_below - conddist);
    const int equal = (int)(first_above - first_not_below);
    if (equal == 0) return (double)below / (double)size;
    return (double)(below + 1 + (equal - 1) / 2.0) / (double)size;
}

inline std::string fmt_g(double v)
{  // default ostream << double (6 significant digits)
    char buf[64];
    snprintf(buf, sizeof buf, "%g", v);
    return buf;
}

// The same text appended to a buffer, for the hundred thousand lines of a report: std::to_chars with chars_format::general
// and precision 6 is specified to produce what printf("%g") produces in the C locale (C++17 [charconv.to.chars]), at less
// than half the cost (80 ns against 190 ns per number here; the report of a 100 k-family table prints 6 million of them).
// cafehost_format_selftest compares the two on random and awkward values (tests/test_host_format.py).
inline void append_g(std::string& out, double v)
{
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::general, 6);
    out.append(buf, r.ptr);
}
inline void append_int(std::string& out, int v)
{  // "%d"
    char buf[16];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    out.append(buf, r.ptr);
}


struct Argument {
    std::string opt;
    std::vector<std::string> argv;
};

inline bool is_number(const std::string& s)
{
    char* end = nullptr;
    strtod(s.c_str(), &end);
    return end != s.c_str() && *end == '\0';
}

// build_argument_list, cafe/cafe_commands.cpp:476-502: "-x" starts an option unless it is a number
inline std::vector<Argument> build_argument_list(const std::vector<std::string>& tokens)
{
    std::vector<Argument> out;
    for (size_t i = 1; i < tokens.size(); ++i) {
        const std::string& t = tokens[i];
        if (t.size() > 1 && t[0] == '-' && !is_number(t)) {
            out.push_back(Argument{t, {}});
        } else if (!out.empty()) {
            out.back().argv.push_back(t);
        }
    }
    return out;
}

}  // namespace cafehost_impl
