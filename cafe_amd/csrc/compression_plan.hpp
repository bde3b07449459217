// compression_plan.hpp -- subtree-state compression: the plan (which nodes become factor tables, their states, tiles and
// levels; the walk's reduced tree and index table).  Built on the host by cafehip_set_families / cafehip_set_tree.
// Part of cafehip.hip (included inside its anonymous namespace; round-5 split).
#pragma once
// ---- subtree-state compression: plan -------------------------------------------------
const cafehip::MfmaSchedule& walk_sched(const cafehip_ctx* c) { return c->walk_compressed ? c->cp.sched : c->msched; }
int walk_cols(const cafehip_ctx* c) { return c->walk_compressed ? c->cp.n_cols : c->n_leaves; }

void free_compression(cafehip_ctx* c)
{
    auto& p = c->cp;
    hipFree(p.d_ops);
    hipFree(p.d_counts);
    hipFree(p.d_col_has_err);
    hipFree(p.d_tiles);
    hipFree(p.d_table_off);
    hipFree(p.d_tables);
    p = cafehip_ctx::CompressPlan();
}

int upload_col_has_err(cafehip_ctx* c)
{
    auto& p = c->cp;
    if (!p.valid) return 0;
    std::vector<uint8_t> v(std::max(p.n_cols, 1), 0);
    for (int j = 0; j < p.n_cols; ++j)
        if (p.col_leaf[j] >= 0 && p.col_leaf[j] < (int)c->h_leaf_has_err.size()) v[j] = c->h_leaf_has_err[p.col_leaf[j]];
    if (!p.d_col_has_err) HIP_TRY(hipMalloc(&p.d_col_has_err, v.size()));
    HIP_TRY(hipMemcpy(p.d_col_has_err, v.data(), v.size(), hipMemcpyHostToDevice));
    return 0;
}

// Paired row tiles for a level of `n_tiles` workgroups (k2c_nodes<.., PAIR>): two row tiles per wave, read as ONE 16-byte load
// per lane and k-step (the even and the odd rows of 32) -- half the vector-memory instructions of the product and half the
// waves per tile, so that four tiles instead of two are resident per CU and one tile's set-up hides behind the others'
// products.  Worth 8-11 % of the table launches where a level is many rounds of tiles (configs[2..4]); a level of a small
// table is ONE round and lasts as long as its slowest wave, which then carries two tiles instead of one (configs[1]: +17 %).
// Option k2c_pair: -1 = by level size (default: at least `k2c_pair_min` tiles per CU), 0 = never, 1 = always.
bool k2c_pairs(const cafehip_ctx* c, long long n_tiles)
{
    const int RT = (c->C + 15) / 16;
    if (RT < 2 || RT > 32 || (c->LD & 1) || c->opt.k2c_pair == 0) return false;
    if (c->opt.k2c_pair > 0) return true;
    return n_tiles >= (long long)c->opt.k2c_pair_min * std::max(c->n_cu, 1);
}

// wave rows / row tiles per wave of k2c_nodes for this matrix side: one wave per row tile up to 16 waves, or one wave per
// PAIR of row tiles (0: matrix too large, no compression)
int k2c_wave_rows(const cafehip_ctx* c, int* nrt_w, bool pair = false)
{
    const int RT = (c->C + 15) / 16;
    if (pair) {
        *nrt_w = 2;
        return (RT + 1) / 2;
    }
    const int wr = std::min(RT, 16);
    *nrt_w = (RT + wr - 1) / wr;
    return *nrt_w <= 2 ? wr : 0;   // (matrix sides up to 512; beyond, the plain walk)
}

// (Re)build the plan from the tree and the unique rows.  A node is compressed when both children are leaves or
// compressed and its distinct states number at most `compress_theta` of the unique rows (default by table size
// and matrix side, see below); option compress=0 disables.  Tables with fewer than 64 unique rows (`compress_min`)
// are left alone.
int rebuild_compression(cafehip_ctx* c, double theta_retry = -1.0)
{
    free_compression(c);
    if (!c->opt.compress) return 0;
    const int n = c->n_nodes, nl = c->n_leaves, Fu = c->Fu;
    const int min_rows = std::max(c->opt.compress_min, 16);   // (default 64) even a 100-row table gains: its walk is a chain of latency-bound steps, and compression shortens the chain
    if (n <= 0 || c->M < 0 || nl != (n + 1) / 2 || Fu < min_rows || (int)c->h_ucounts.size() != Fu * nl) return 0;
    int nrt_w = 0;
    if (k2c_wave_rows(c, &nrt_w) == 0) return 0;
    const auto& left = c->left;
    const auto& right = c->right;
    auto internal = [&](int v) { return left[v] >= 0; };
    std::vector<int> post;
    {
        std::vector<std::pair<int, int>> st;
        st.push_back({c->root, 0});
        while (!st.empty()) {
            auto& top = st.back();
            const int v = top.first;
            if (!internal(v)) { post.push_back(v); st.pop_back(); }
            else if (top.second == 0) { top.second = 1; st.push_back({left[v], 0}); }
            else if (top.second == 1) { top.second = 2; st.push_back({right[v], 0}); }
            else { post.push_back(v); st.pop_back(); }
        }
    }
    // Threshold.  A table product costs more per state than a walk product per family (16-state tiles re-read the
    // matrix: x1.5 at a 151-wide matrix, x1.2 at 251) and every level is a launch: for tables that fill the chip
    // the measured optimum is 0.5 / 0.7 (sweep of 0.2..0.9 at the bench shapes).  A SMALL table does not fill the
    // chip either way; its cost is the length of the dependency chain -- one latency-bound step of the walk per
    // internal node against one launch per LEVEL of compressed nodes, all nodes of a level side by side -- so
    // everything below the root is "compressed" whatever the number of states (250..2,000 rows on the 32- and
    // 64-taxon trees: 1.4-2.5x faster than 0.5), with 0.8 in between (sweeps at 250..10,000 rows).
    double theta = c->C < 200 ? 0.5 : 0.7;
    {
        // (a launch costs about two walk steps: the small-table rule only where the tree is at most half as deep as
        // it has internal nodes -- not for a caterpillar, whose every node is a level of its own)
        std::vector<int> height(n, 0);
        int n_internal = 0;
        for (int v : post)
            if (left[v] >= 0) {
                height[v] = 1 + std::max(height[left[v]], height[right[v]]);
                ++n_internal;
            }
        const bool bushy = 2 * (height[c->root] - 1) <= n_internal - 1;
        if (bushy && Fu < 10 * std::max(c->n_cu, 1)) theta = 1.0;
        else if (bushy && Fu < 32 * std::max(c->n_cu, 1)) theta = 0.8;
        // round 5, with the paired table kernel: a LARGE table on a narrow matrix is served as well by 0.7 as a wide one
        // (151-wide, 160 k rows: 0.663 -> 0.644 ms; 62.5 k rows on 64 taxa: 1.444 -> 1.438; 40 k rows: the same plan either
        // way; profiles/r05/theta_sweep2_paired_tables.txt)
        else if (c->C < 200 && Fu >= 160 * std::max(c->n_cu, 1)) theta = 0.7;
        // round 6, with k2c_gemm: on a WIDE matrix (16 row tiles: 32-state tiles, 0.77 of the matrix peak, no parked vectors) a
        // table product is cheaper than a walk product even where nothing is shared, so a large table takes EVERY node below
        // the root as a table and the walk is the root step alone: configs[2] 1.951 -> 1.856 ms (0.9: 1.892), configs[4]
        // 1.990 -> 1.900; on a 151-wide matrix (16-state tiles) it is 14 % slower and the threshold stays
        // (profiles/r06/theta_sweep_k2c_gemm.txt).  The tables then grow to (internal nodes) x rows x LD doubles at most:
        // kept below 8 GiB per parameter set and below the 2^31-element offsets, else the plan is rebuilt with 0.7.
        // (from 8 k rows on: 10 k rows 0.392 -> 0.365 ms, 20 k 0.624 -> 0.533, 40 k 0.949 -> 0.888, 100 k 1.941 -> 1.845)
        else if (c->C >= 200 && Fu >= 32 * std::max(c->n_cu, 1) && c->opt.k2c_gemm != 0) theta = 1.0;
    }
    if (theta_retry >= 0) theta = theta_retry;
    else if (c->opt.compress_theta >= 0) theta = std::min(c->opt.compress_theta, 1.0);
    const size_t limit = (size_t)(theta * Fu);
    const int max_level = c->opt.compress_max_level > 0 ? c->opt.compress_max_level : INT_MAX;
    // a table whose walk is one round of workgroups (at most 64 rows each): an evaluation is a chain of launches and
    // latency-bound steps, not work
    const bool launch_bound = Fu <= 64 * std::max(c->n_cu, 1);
    std::vector<std::vector<int32_t>> sid(n), idx0(n), idx1(n);
    std::vector<int> D(n, 0), level(n, 0);
    std::vector<char> comp(n, 0);
    // One node's states: numbered in order of first appearance over the unique rows (deterministic, whatever runs
    // beside it).  A node needs only its two children, and a child has one parent: the nodes of one height are
    // planned side by side on host threads (100 k rows, 32 taxa: 29 -> ~8 ms of set-up).
    auto plan_node = [&](int v) {
        if (!internal(v)) {
            sid[v].resize(Fu);
            for (int u = 0; u < Fu; ++u) sid[v][u] = c->h_ucounts[(size_t)u * nl + v / 2];
            return;
        }
        const int a = left[v], b = right[v];
        const bool ok_children = (!internal(a) || comp[a]) && (!internal(b) || comp[b]);
        if (v != c->root && ok_children) {
            // open-addressing table keyed by the pair of child states (linear probing; at most `limit` entries in a power
            // of two of at least twice that)
            size_t cap = 64;
            int cap_log2 = 6;
            while (cap < 2 * (std::min<size_t>(limit, (size_t)Fu) + 1)) { cap <<= 1; ++cap_log2; }
            std::vector<uint64_t> keys(cap);
            std::vector<int32_t> vals(cap, -1);
            int32_t n_ids = 0;
            std::vector<int32_t> mine(Fu);
            bool fits = true;
            const int32_t *sa = sid[a].data(), *sb = sid[b].data();
            for (int u = 0; u < Fu; ++u) {
                const uint64_t key = ((uint64_t)(uint32_t)sa[u] << 32) | (uint32_t)sb[u];
                // home slot from the TOP bits of the product: both children's states reach them (the left child's state sits
                // in the key's high half and only enters the product's bits from 32 up)
                size_t at = (size_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - cap_log2));
                while (vals[at] >= 0 && keys[at] != key) at = (at + 1) & (cap - 1);
                if (vals[at] < 0) {
                    if ((size_t)n_ids >= limit) { fits = false; break; }
                    keys[at] = key;
                    vals[at] = n_ids++;
                    idx0[v].push_back(sa[u]);
                    idx1[v].push_back(sb[u]);
                }
                mine[u] = vals[at];
            }
            const int lvl = 1 + std::max(comp[a] ? level[a] : 0, comp[b] ? level[b] : 0);
            if (fits && n_ids > 0 && lvl <= max_level) {
                comp[v] = 1;
                D[v] = n_ids;
                level[v] = lvl;
                sid[v].swap(mine);
            } else {
                idx0[v].clear();
                idx1[v].clear();
            }
        }
        // the children's states are needed again only as columns of the walk (kept below for the maximal nodes; a small
        // table keeps them all: its top levels may go back to the walk, see below)
        if (comp[v] && !launch_bound) {
            std::vector<int32_t>().swap(sid[a]);
            std::vector<int32_t>().swap(sid[b]);
        }
    };
    {
        std::vector<int> height(n, 0);
        int top = 0;
        for (int v : post)
            if (internal(v)) {
                height[v] = 1 + std::max(height[left[v]], height[right[v]]);
                top = std::max(top, height[v]);
            }
        for (int h = 0; h <= top; ++h) {
            std::vector<int> wave;
            for (int v : post)
                if (height[v] == h) wave.push_back(v);
            const int workers = std::min<int>({(int)wave.size(), 16, std::max(1, (int)std::thread::hardware_concurrency())});
            if (workers <= 1 || Fu < 32768) {
                for (int v : wave) plan_node(v);
                continue;
            }
            std::atomic<size_t> next{0};
            std::vector<std::thread> pool;
            for (int w = 0; w < workers; ++w)
                pool.emplace_back([&] {
                    for (size_t i = next++; i < wave.size(); i = next++) plan_node(wave[i]);
                });
            for (auto& th : pool) th.join();
        }
    }
    std::vector<int> parent(n, -1);
    for (int v = 0; v < n; ++v)
        if (internal(v)) { parent[left[v]] = v; parent[right[v]] = v; }
    int n_comp = 0, n_levels = 0;
    for (int v = 0; v < n; ++v)
        if (comp[v]) { ++n_comp; n_levels = std::max(n_levels, level[v]); }
    // Predicted TIME, not work, decides the top of the forest of a launch-bound table (round 5).  There a level costs its
    // launch -- ~8 us with the gap in front of it, whatever its tile count up to a chip-full (the reference's test1 table:
    // 6.5-7.7 us for 73-367 tiles) -- while a node left to the walk costs one more walk step: ~2.5 us of gathers and
    // barriers + the product, 0.2 us per k-step at ten row tiles (10 us at a 151-wide matrix, 4-5 us at 71).  A top level
    // whose nodes are cheaper as walk steps goes back to the walk; measured on test1 (profiles/r05/plan_sweep.txt): 81.9 us
    // with all five levels, 79.0 without the fifth, 77.5 without the fourth too, 80.9 once the two nodes of the third go.
    if (launch_bound && c->opt.compress_drop_top && c->opt.compress_max_level <= 0) {
        const double ksteps = (c->C + 3) / 4, row_tiles = (c->C + 15) / 16;
        const double step_us = 2.5 + 0.2 * ksteps * row_tiles / 10.0, level_us = 8.0;
        while (n_levels >= 2) {
            int n_top = 0;
            for (int v = 0; v < n; ++v)
                if (comp[v] && level[v] == n_levels) ++n_top;
            if (n_top * step_us + 0.5 >= level_us) break;
            for (int v = 0; v < n; ++v)
                if (comp[v] && level[v] == n_levels) {
                    comp[v] = 0;
                    D[v] = 0;
                    level[v] = 0;
                    std::vector<int32_t>().swap(idx0[v]);
                    std::vector<int32_t>().swap(idx1[v]);
                    std::vector<int32_t>().swap(sid[v]);
                    --n_comp;
                }
            --n_levels;
        }
    }
    if (n_comp == 0) return 0;
    auto& p = c->cp;
    // tables
    std::vector<int32_t> table_off(n, 0);
    size_t elems = 0, n_idx = 0;
    for (int v = 0; v < n; ++v)
        if (comp[v]) {
            table_off[v] = (int32_t)elems;
            elems += (size_t)D[v] * c->LD;
            n_idx += 2 * (size_t)D[v];
            p.states += D[v];
        }
    if (elems >= ((size_t)1 << 31) || n_idx >= ((size_t)1 << 31) || elems * sizeof(double) > ((size_t)8 << 30)) {
        p = cafehip_ctx::CompressPlan();
        // (too many states for one allocation / the 32-bit offsets: once more with the threshold that shares, then not at all)
        if (theta > 0.7 && theta_retry < 0) return rebuild_compression(c, 0.7);
        return 0;
    }
    p.table_elems = elems;
    // tiles, level by level (children's tables are complete before a level starts)
    p.level_first.assign(1, 0);
    p.level_nft.clear();
    p.level_nrt.clear();
    for (int l = 1; l <= n_levels; ++l) {
        // k2c_nodes: 16 states per tile (32- and 64-state tiles of THAT kernel -- a half / a quarter of the workgroups and of
        // the matrix re-reads, but the whole node vectors gathered up front -- measured 2-30 % slower at every bench shape).
        // k2c_gemm (round 6): 16 * nft states per tile with the node vectors formed chunk by chunk; large tiles only where the
        // level still gives every CU a couple of them.
        int nft = 1, nrt = 0;
        {
            long long states = 0, tiles16 = 0;
            for (int v = 0; v < n; ++v)
                if (comp[v] && level[v] == l) { states += D[v]; tiles16 += (D[v] + 15) / 16; }
            const int RT = (c->C + 15) / 16, n_cu = std::max(c->n_cu, 1);
            const bool gemm = c->opt.k2c_gemm > 0 || c->opt.k2c_gemm < 0;
            if (gemm && RT <= 32 && !(c->LD & 1)) {
                const bool pair = RT >= 2 && (RT > 16 || c->opt.k2c_pair > 0 || (c->opt.k2c_pair < 0 && tiles16 >= (long long)c->opt.k2c_pair_min * n_cu));
                nrt = pair ? 2 : 1;
                const int waves = (RT + nrt - 1) / nrt;
                // Measured (profiles/r06/k2c_gemm_ab.txt): what pays is a tile whose gathers are ONE 16-byte slot per thread
                // (256 * nft slots over 64 * waves threads) -- 32-state tiles on a 251-wide matrix (8 waves: -10 % against
                // k2c_nodes), 16-state tiles on a 151-wide one (5 waves: -9 %).  A second slot per thread costs the registers
                // of a resident wave per SIMD, and 64-state tiles measured 4-30 % SLOWER than 16-state ones everywhere: a
                // level is bound by the tiles in flight per CU, not by the matrix operand's bytes.
                nft = c->opt.k2c_nst > 0 ? c->opt.k2c_nst : ((states >= 64LL * n_cu && waves >= 8) ? 2 : 1);
                while (nft > waves) nft >>= 1;   // (at most 4 gather slots per thread)
                nft = std::max(nft, 1);
            }
        }
        const int ts = 16 * nft;
        for (int v = 0; v < n; ++v) {
            if (!comp[v] || level[v] != l) continue;
            ++p.n_nodes;
            const int ch[2] = {left[v], right[v]};
            for (int s0 = 0; s0 < D[v]; s0 += ts) {
                cafehip::CTile t{};
                t.node = v;
                t.state0 = s0;
                t.n_live = std::min(ts, D[v] - s0);
                t.out_off = table_off[v];
                for (int k = 0; k < 2; ++k) {
                    t.child[k] = ch[k];
                    t.kind[k] = internal(ch[k]) ? 2 : 0;
                    t.leafcol[k] = internal(ch[k]) ? 0 : ch[k] / 2;
                    t.tab_off[k] = internal(ch[k]) ? table_off[ch[k]] : 0;
                    const auto& ix = k ? idx1[v] : idx0[v];
                    for (int f = 0; f < t.n_live; ++f) t.idx[k][f] = ix[s0 + f];
                }
                p.tiles.push_back(t);
            }
        }
        p.level_first.push_back((int)p.tiles.size());
        p.level_nft.push_back(nft);
        p.level_nrt.push_back(nrt);
    }
    // the reduced tree's leaves and the walk's index table
    std::vector<char> under(n, 0);   // strictly below a compressed node
    for (int i = (int)post.size() - 1; i >= 0; --i) {
        const int v = post[i];   // parents before children in reverse post-order
        if (parent[v] >= 0 && (comp[parent[v]] || under[parent[v]])) under[v] = 1;
    }
    std::vector<int> leafcol_of(n, -1);
    for (int v = 0; v < n; ++v) {
        if (under[v]) continue;
        if (!internal(v) || comp[v]) {
            if (internal(v)) p.top_states += D[v];
            leafcol_of[v] = p.n_cols++;
            p.col_leaf.push_back(internal(v) ? -1 : v / 2);
        }
    }
    std::vector<int32_t> wc((size_t)Fu * p.n_cols);
    for (int v = 0; v < n; ++v) {
        const int j = leafcol_of[v];
        if (j < 0) continue;
        if (internal(v))
            for (int u = 0; u < Fu; ++u) wc[(size_t)u * p.n_cols + j] = sid[v][u];
        else
            for (int u = 0; u < Fu; ++u) wc[(size_t)u * p.n_cols + j] = c->h_ucounts[(size_t)u * nl + v / 2];
    }
    p.sched = cafehip::build_mfma_schedule(n, c->root, left, right, &leafcol_of);
    HIP_TRY(hipMalloc(&p.d_ops, std::max<size_t>(p.sched.ops.size(), 1) * sizeof(cafehip::MfmaOp)));
    HIP_TRY(hipMemcpy(p.d_ops, p.sched.ops.data(), p.sched.ops.size() * sizeof(cafehip::MfmaOp), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p.d_counts, wc.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p.d_counts, wc.data(), wc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p.d_tiles, p.tiles.size() * sizeof(cafehip::CTile)));
    HIP_TRY(hipMemcpy(p.d_tiles, p.tiles.data(), p.tiles.size() * sizeof(cafehip::CTile), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p.d_table_off, n * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p.d_table_off, table_off.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    p.valid = true;
    return upload_col_has_err(c);
}

