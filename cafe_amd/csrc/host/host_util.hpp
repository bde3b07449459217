// host_util.hpp -- number formatting as the reference prints it, the rank of a likelihood in a sorted null, command-line argument lists
// (part of the host driver, cafe_host.cpp; split out in round 4 so that the session file holds the commands only)
#pragma once
#include <algorithm>
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
