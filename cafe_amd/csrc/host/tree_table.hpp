// tree_table.hpp -- the tree (Newick -> nlist arrays), the family table and the size ranges as the reference's load / tree commands leave them
// (part of the host driver, cafe_host.cpp; split out in round 4 so that the session file holds the commands only)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cafehost_impl {

// ------------------------------------------------------------------------------------
// tree: Newick -> nlist arrays (tree_build_node_list, cafe/cafe_commands.cpp:2028-2051)
// ------------------------------------------------------------------------------------
struct HostTree {
    int n = 0, root = -1;
    std::vector<int32_t> parent, left, right;
    std::vector<double> bl;
    std::vector<std::string> name;
    std::string newick;  // as typed

    int n_leaves() const { return (n + 1) / 2; }

    static HostTree parse(const std::string& text)
    {
        std::string s = text;
        while (!s.empty() && (s.back() == ';' || isspace((unsigned char)s.back()))) s.pop_back();
        struct Raw {
            std::string name;
            double bl = -1.0;  // the root keeps -1 (libtree/phylogeny.c)
            std::vector<int> kids;
        };
        std::vector<Raw> raw;
        size_t pos = 0;
        std::function<int()> rec = [&]() -> int {
            const int me = (int)raw.size();
            raw.emplace_back();
            if (pos < s.size() && s[pos] == '(') {
                ++pos;
                while (true) {
                    const int ch = rec();
                    raw[me].kids.push_back(ch);
                    if (pos < s.size() && s[pos] == ',') {
                        ++pos;
                        continue;
                    }
                    if (pos >= s.size() || s[pos] != ')') throw std::runtime_error("Failed to load tree from provided string");
                    ++pos;
                    break;
                }
            }
            size_t j = pos;
            while (j < s.size() && s[j] != ',' && s[j] != '(' && s[j] != ')' && s[j] != ':') ++j;
            raw[me].name = s.substr(pos, j - pos);
            pos = j;
            if (pos < s.size() && s[pos] == ':') {
                size_t k = pos + 1;
                while (k < s.size() && s[k] != ',' && s[k] != '(' && s[k] != ')') ++k;
                raw[me].bl = atof(s.substr(pos + 1, k - pos - 1).c_str());
                pos = k;
            }
            return me;
        };
        const int r = rec();
        if (pos != s.size()) throw std::runtime_error("Failed to load tree from provided string");
        // in-order numbering: even = leaf, odd = internal
        std::vector<int> order;
        std::vector<std::pair<int, int>> st;
        st.push_back({r, 0});
        while (!st.empty()) {
            auto [v, stage] = st.back();
            st.pop_back();
            if (raw[v].kids.empty()) {
                order.push_back(v);
            } else if (stage == 0) {
                if (raw[v].kids.size() != 2) throw std::runtime_error("Tree must be binary");
                st.push_back({v, 1});
                st.push_back({raw[v].kids[0], 0});
            } else {
                order.push_back(v);
                st.push_back({raw[v].kids[1], 0});
            }
        }
        HostTree t;
        t.n = (int)order.size();
        std::vector<int> id(raw.size());
        for (int i = 0; i < t.n; ++i) id[order[i]] = i;
        t.parent.assign(t.n, -1);
        t.left.assign(t.n, -1);
        t.right.assign(t.n, -1);
        t.bl.assign(t.n, -1.0);
        t.name.assign(t.n, "");
        for (size_t v = 0; v < raw.size(); ++v) {
            const int i = id[v];
            t.name[i] = raw[v].name;
            t.bl[i] = raw[v].bl;
            if (!raw[v].kids.empty()) {
                t.left[i] = id[raw[v].kids[0]];
                t.right[i] = id[raw[v].kids[1]];
                t.parent[t.left[i]] = i;
                t.parent[t.right[i]] = i;
            }
        }
        t.root = id[r];
        t.newick = s;
        return t;
    }

    double max_branch_length() const
    {
        double m = 0;
        for (double b : bl) m = std::max(m, b);
        return m;
    }
};

inline bool iequals(const std::string& a, const std::string& b)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
    return true;
}

// ------------------------------------------------------------------------------------
// family table (load_gene_families, cafe/gene_family.cpp:186-225)
// ------------------------------------------------------------------------------------
struct HostFamilies {
    std::string path;
    std::vector<std::string> species, ids, desc;
    std::vector<int32_t> counts;  // F x species.size(), file column order
    int max_size = 0;
    int F() const { return (int)ids.size(); }

    static std::vector<std::string> split(const std::string& s, char sep)
    {
        std::vector<std::string> out;
        std::string cur;
        for (char ch : s) {
            if (ch == sep) {
                out.push_back(cur);
                cur.clear();
            } else {
                cur.push_back(ch);
            }
        }
        out.push_back(cur);
        return out;
    }

    void load(const std::string& file, int max_size_filter)
    {
        std::ifstream in(file);
        if (!in) throw std::runtime_error("ERROR(load): Cannot open " + file + " in read mode.");
        path = file;
        std::string line;
        if (!std::getline(in, line)) throw std::runtime_error("Failed to identify species for gene families");
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        const char sep = (line.find('\t') != std::string::npos) ? '\t' : ',';
        auto head = split(line, sep);
        if (head.size() < 3) throw std::runtime_error("Failed to identify species for gene families");
        species.assign(head.begin() + 2, head.end());
        ids.clear();
        desc.clear();
        counts.clear();
        max_size = 0;
        while (std::getline(in, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            if (line.empty()) continue;
            auto v = split(line, sep);
            if (v.size() != species.size() + 2)
                throw std::runtime_error("Inconsistency in column count: expected " + std::to_string(species.size() + 2) +
                                         ", but found " + std::to_string(v.size()));
            std::vector<int32_t> row(species.size());
            int mx = 0;
            for (size_t i = 0; i < species.size(); ++i) {
                char* end = nullptr;
                const long val = strtol(v[i + 2].c_str(), &end, 10);
                if (end == v[i + 2].c_str()) throw std::runtime_error("Error reading family '" + v[1] + "'");
                row[i] = (int32_t)val;
                mx = std::max(mx, (int)val);
            }
            // cafe/gene_family.cpp:217: keep the row when max_size < 0 or max(row) <= max_size
            if (max_size_filter < 0 || mx <= max_size_filter) {
                desc.push_back(v[0]);
                ids.push_back(v[1]);
                counts.insert(counts.end(), row.begin(), row.end());
                max_size = std::max(max_size, mx);
            }
        }
    }
};

struct HostRange {
    int min = 0, max = 0, root_min = 1, root_max = 1;
};

// init_family_size, cafe/cafe_family.c:357-364
inline HostRange init_family_size(int max)
{
    HostRange r;
    r.root_min = 1;
    r.root_max = (int)std::max(30.0, std::rint(max * 1.25));
    r.max = max + std::max(50, max / 5);
    r.min = 0;
    return r;
}

}  // namespace cafehost_impl
