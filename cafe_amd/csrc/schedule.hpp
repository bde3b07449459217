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


// ---------------------------------------------------------------------------------
// Schedule for the MFMA kernel: ONE node-vector buffer in LDS per workgroup.  A node's
// result stays in the buffer when the very next step consumes it (its parent), and is
// "parked" in a per-workgroup global scratch region otherwise (the next step starts the
// sibling subtree).  Children are ordered so that the LDS-resident one is multiplied first.
// Visiting the child whose subtree needs more parks first keeps the park count at the
// Sethi-Ullman minimum (<= log2 #internal nodes).
// ---------------------------------------------------------------------------------
struct MfmaOp {
    int32_t node;
    int32_t is_root;
    int32_t child[2];     // child node ids, in processing order
    int32_t kind[2];      // 0 = leaf (column gather), 1 = internal (vector in LDS or parked),
                          // 2 = compressed subtree (row gather from its factor table, see CNode)
    int32_t leafcol[2];   // count-table column for leaves / state-id column for compressed subtrees
    int32_t src_park[2];  // internal child: -1 = vector is in the LDS buffer, else park index
    int32_t dst_park;     // -1 = keep the result in the LDS buffer, else park index
    int32_t pad;
};

struct MfmaSchedule {
    std::vector<MfmaOp> ops;
    int n_parks = 0;
};

// `leafcol_of` (optional): the walk of a REDUCED tree -- a node v with leafcol_of[v] >= 0 is a leaf of the walk
// whatever its children: an original leaf (kind 0) or the root of a compressed subtree (kind 2), its index column
// in the walk's count table being leafcol_of[v].  Without it the leaves are the tree's (column = node id / 2).
inline MfmaSchedule build_mfma_schedule(int n_nodes, int root, const std::vector<int>& left_in,
                                        const std::vector<int>& right_in, const std::vector<int>* leafcol_of = nullptr)
{
    MfmaSchedule sch;
    std::vector<int> left = left_in, right = right_in;
    if (leafcol_of)
        for (int v = 0; v < n_nodes; ++v)
            if ((*leafcol_of)[v] >= 0) left[v] = right[v] = -1;
    // parks needed by each subtree
    std::vector<int> need(n_nodes, 0);
    std::vector<int> post;
    {
        std::vector<std::pair<int, int>> st;
        st.push_back({root, 0});
        while (!st.empty()) {
            auto& top = st.back();
            const int v = top.first;
            if (left[v] < 0) {
                post.push_back(v);
                st.pop_back();
            } else if (top.second == 0) {
                top.second = 1;
                st.push_back({left[v], 0});
            } else if (top.second == 1) {
                top.second = 2;
                st.push_back({right[v], 0});
            } else {
                post.push_back(v);
                st.pop_back();
            }
        }
    }
    auto internal = [&](int v) { return left[v] >= 0; };
    for (int v : post) {
        if (!internal(v)) continue;
        const int a = left[v], b = right[v];
        if (internal(a) && internal(b)) {
            const int hi = std::max(need[a], need[b]), lo = std::min(need[a], need[b]);
            need[v] = std::max(hi, lo + 1);
        } else if (internal(a) || internal(b)) {
            need[v] = internal(a) ? need[a] : need[b];
        } else {
            need[v] = 0;
        }
    }
    // emission order (iterative DFS, larger-need internal child first)
    std::vector<int> order;
    {
        struct Fr { int v, stage, first, second; };
        std::vector<Fr> st;
        st.push_back({root, 0, -1, -1});
        while (!st.empty()) {
            Fr f = st.back();
            const int v = f.v;
            if (!internal(v)) { st.pop_back(); continue; }
            if (f.stage == 0) {
                int a = left[v], b = right[v];
                // internal before leaf is irrelevant (leaves emit nothing); among two internal
                // children take the larger need first
                if (internal(a) && internal(b) && need[b] > need[a]) std::swap(a, b);
                st.back().stage = 1;
                st.back().first = a;
                st.back().second = b;
                st.push_back({a, 0, -1, -1});
            } else if (f.stage == 1) {
                st.back().stage = 2;
                st.push_back({f.second, 0, -1, -1});
            } else {
                order.push_back(v);
                st.pop_back();
            }
        }
    }
    std::vector<int> parent(n_nodes, -1);
    for (int v = 0; v < n_nodes; ++v)
        if (internal(v)) { parent[left[v]] = v; parent[right[v]] = v; }
    std::vector<int> park_of(n_nodes, -1);
    std::vector<int> free_parks;
    int next_park = 0;
    int in_lds = -1;  // node whose vector currently sits in the LDS buffer
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const int v = order[oi];
        MfmaOp op{};
        op.node = v;
        op.is_root = (v == root);
        int ch[2] = {left[v], right[v]};
        // LDS-resident internal child first
        if (internal(ch[1]) && ch[1] == in_lds) std::swap(ch[0], ch[1]);
        for (int c = 0; c < 2; ++c) {
            op.child[c] = ch[c];
            op.src_park[c] = -1;
            op.leafcol[c] = 0;
            if (!internal(ch[c])) {
                op.kind[c] = (left_in[ch[c]] < 0) ? 0 : 2;
                op.leafcol[c] = leafcol_of ? (*leafcol_of)[ch[c]] : ch[c] / 2;
            } else {
                op.kind[c] = 1;
                if (ch[c] != in_lds) {
                    op.src_park[c] = park_of[ch[c]];
                    free_parks.push_back(park_of[ch[c]]);
                    park_of[ch[c]] = -1;
                }
            }
        }
        const bool next_is_parent = (oi + 1 < order.size()) && (order[oi + 1] == parent[v]);
        if (v == root || next_is_parent) {
            op.dst_park = -1;
            in_lds = v;
        } else {
            int p;
            if (!free_parks.empty()) { p = free_parks.back(); free_parks.pop_back(); }
            else p = next_park++;
            op.dst_park = p;
            park_of[v] = p;
            // in_lds unchanged only if this op did not touch the buffer; a parked result never
            // lands in LDS, but fetching a parked child overwrote the buffer:
            if (op.src_park[0] >= 0 || op.src_park[1] >= 0) in_lds = -1;
            else if (in_lds == ch[0] || in_lds == ch[1]) in_lds = -1;  // consumed
        }
        sch.ops.push_back(op);
    }
    sch.n_parks = next_park;
    return sch;
}


// ---------------------------------------------------------------------------------
// Subtree-state compression.  The vector of an internal node depends on the family only through the counts at the
// leaves below it; families that agree there share the vector, and its product with the node's edge matrix.  A
// COMPRESSED node v has D_v distinct states (tuples of its children's states; a leaf's state is its count) among
// the table's unique rows, few enough that the factor M_v . L_v is built once per state into a table
// [D_v][LD] (k2c_nodes: tiles of 16 states, one launch per level, children before parents) and the family walk gathers row
// state_v(family) from it exactly as it gathers a matrix column for a one-hot leaf.  Same products on the same
// operands in the same order: bit-identical to the uncompressed walk.
// ---------------------------------------------------------------------------------
// One workgroup's work in k2c_nodes, self-contained (one load instead of a tile -> node -> index chain of dependent
// global round trips: a level is latency-bound): up to kCTileStates states of one compressed node.
constexpr int kCTileStates = 64;
struct CTile {
    int32_t node;        // tree node whose edge matrix multiplies the vectors
    int32_t state0;      // first state of the tile
    int32_t n_live;      // states in the tile (1 .. 16 * NFT_W of the level's launch)
    int32_t out_off;     // element offset of the node's table
    int32_t child[2];
    int32_t kind[2];     // 0 = leaf, 2 = compressed child
    int32_t leafcol[2];
    int32_t tab_off[2];
    int32_t pad[4];
    int32_t idx[2][kCTileStates];  // per state of the tile: the children's counts / states
};

}  // namespace cafehip
