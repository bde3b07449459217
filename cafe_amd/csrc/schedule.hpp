// schedule.hpp -- static evaluation order of the pruning recursion.
//
// The reference walks the tree recursively, left subtree, right subtree, node
// (cafe/cafe_tree.c:301-318) with a full likelihood vector per node.  On the GPU a
// tile of families keeps its node vectors in LDS, so the order is chosen to keep as
// few vectors alive as possible (Sethi-Ullman numbering; leaves cost nothing because
// a one-hot leaf turns the edge product into a column gather, cafe/cafe_tree.c:208-209).
// L_v = (P_a L_a) .* (P_b L_b) is symmetric in the children, so the visiting order
// does not change any value.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace cafehip {

struct PruneOp {
    int32_t node;      // internal node produced by this step
    int32_t dst;       // LDS slot receiving L_node (may equal a source slot: in-place)
    int32_t is_root;   // rows = root range instead of [range_min, range_max]
    int32_t child[2];  // child node ids (edge = child's branch)
    int32_t kind[2];   // 0 = leaf (src = leaf column in the count table), 1 = slot
    int32_t src[2];
};

struct Schedule {
    std::vector<PruneOp> ops;
    int n_slots = 0;
};

inline Schedule build_schedule(int n_nodes, int root, const std::vector<int>& left,
                               const std::vector<int>& right)
{
    Schedule sch;
    std::vector<int> need(n_nodes, 0);
    // post-order without recursion (trees can be caterpillars)
    std::vector<int> order;
    order.reserve(n_nodes);
    {
        std::vector<std::pair<int, int>> st;
        st.push_back({root, 0});
        while (!st.empty()) {
            auto& top = st.back();
            int v = top.first;
            if (left[v] < 0) {
                order.push_back(v);
                st.pop_back();
            } else if (top.second == 0) {
                top.second = 1;
                st.push_back({left[v], 0});
            } else if (top.second == 1) {
                top.second = 2;
                st.push_back({right[v], 0});
            } else {
                order.push_back(v);
                st.pop_back();
            }
        }
    }
    for (int v : order) {
        if (left[v] < 0) {
            need[v] = 0;
            continue;
        }
        int a = need[left[v]], b = need[right[v]];
        bool ia = left[left[v]] >= 0, ib = left[right[v]] >= 0;
        if (ia && ib)
            need[v] = (a == b) ? a + 1 : std::max(a, b);
        else if (ia || ib)
            need[v] = std::max(ia ? a : b, 1);
        else
            need[v] = 1;
    }
    std::vector<int> free_slots;
    int next_slot = 0;
    auto alloc = [&]() {
        if (!free_slots.empty()) {
            int s = free_slots.back();
            free_slots.pop_back();
            return s;
        }
        return next_slot++;
    };
    std::vector<int> slot_of(n_nodes, -1);
    // explicit stack: visit the child with the larger need first
    struct Frame {
        int v, stage, first, second;
    };
    std::vector<Frame> st;
    st.push_back({root, 0, -1, -1});
    while (!st.empty()) {
        Frame& f = st.back();
        int v = f.v;
        if (left[v] < 0) {
            st.pop_back();
            continue;
        }
        if (f.stage == 0) {
            int a = left[v], b = right[v];
            if (need[b] > need[a]) std::swap(a, b);
            f.first = a;
            f.second = b;
            f.stage = 1;
            st.push_back({a, 0, -1, -1});
            continue;
        }
        if (f.stage == 1) {
            f.stage = 2;
            st.push_back({f.second, 0, -1, -1});
            continue;
        }
        PruneOp op{};
        op.node = v;
        op.is_root = (v == root) ? 1 : 0;
        op.child[0] = left[v];
        op.child[1] = right[v];
        int dst = -1;
        for (int c = 0; c < 2; ++c) {
            int ch = op.child[c];
            if (left[ch] < 0) {
                op.kind[c] = 0;
                op.src[c] = ch / 2;  // leaf node id 2j <-> count column j
            } else {
                op.kind[c] = 1;
                op.src[c] = slot_of[ch];
                if (dst < 0)
                    dst = slot_of[ch];
                else
                    free_slots.push_back(slot_of[ch]);
            }
        }
        if (dst < 0) dst = alloc();
        op.dst = dst;
        slot_of[v] = dst;
        sch.ops.push_back(op);
        st.pop_back();
    }
    sch.n_slots = std::max(next_slot, 1);
    return sch;
}

}  // namespace cafehip
