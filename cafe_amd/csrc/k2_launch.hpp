// k2_launch.hpp -- the pruning launches of an evaluation: factor-table levels (k2c_nodes), the matrix-core walk with its park
// scratch, wave-grid cost model and measured choice, the row-per-thread fallback's dispatcher.
// Part of cafehip.hip (included inside its anonymous namespace; round-5 split).
#pragma once
int launch_k2c_inst(cafehip_ctx* c, const void* fn, int nft_w, const K2cArgs& a_in, int grid, int n_sets, int block)
{
    K2cArgs a = a_in;
    a.block_threads = block;
    const size_t lds = (size_t)16 * nft_w * c->LDv * sizeof(double);
    if (!fn) return fail("internal: no k2c_nodes instantiation for this shape");
    if (grant_lds(c, fn, lds, 64 * 1024)) return -1;
#ifdef CAFE_K2_STAMPS
    if (const char* stamps_file = getenv("CAFEHIP_STAMPS_FILE")) {
        // debug builds: per (tile, wave) s_memtime stamps of this level, appended to <file>.k2c
        K2cArgs b = a;
        const size_t n = (size_t)grid * 16 * 8;
        unsigned long long* d = nullptr;
        HIP_TRY(hipMalloc(&d, n * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d, 0, n * sizeof(unsigned long long), c->stream));
        b.stamps = d;
        if (launch_kernel(fn, dim3(grid, n_sets), dim3(block), lds, c->stream, b)) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<unsigned long long> h(n);
        HIP_TRY(hipMemcpy(h.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        hipFree(d);
        if (FILE* f = fopen((std::string(stamps_file) + ".k2c").c_str(), "ab")) {
            const long long hdr[4] = {grid, block / 64, 8, 0};
            fwrite(hdr, sizeof hdr, 1, f);
            fwrite(h.data(), sizeof(unsigned long long), n, f);
            fclose(f);
        }
        return 0;
    }
#endif
    return launch_kernel(fn, dim3(grid, n_sets), dim3(block), lds, c->stream, a);
}

// factor tables of the compressed subtrees for the matrices just built: one launch per level, children first
int launch_compressed_levels(cafehip_ctx* c, int n_sets)
{
    auto& p = c->cp;
    c->issued_tables = 0;
    if (!p.valid) return 0;
    const size_t need = p.table_elems * (size_t)n_sets;
    if (need > p.tables_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(p.d_tables);
        p.d_tables = nullptr;
        p.tables_cap = 0;
        HIP_TRY(hipMalloc(&p.d_tables, need * sizeof(double)));
        HIP_TRY(hipMemsetAsync(p.d_tables, 0, need * sizeof(double), c->stream));   // row padding beyond the tiles stays zero
        p.tables_cap = need;
    }
    K2cArgs a;
    memset(&a, 0, sizeof a);
    a.PT = c->d_PT;
    a.PTfold = (c->d_err && c->fold_current) ? c->d_PTfold : nullptr;
    a.node_key = c->cur_node_key;
    a.n_nodes = c->n_nodes;
    a.leaf_has_err32 = c->d_leaf_has_err32;
    a.tables = p.d_tables;
    a.table_set_stride = p.table_elems;
    a.C = c->C;
    a.LD = c->LD;
    a.KP = c->KP;
    a.LDv = c->LDv;
    a.ksteps = (c->C + 3) / 4;
    double slots = 0;   // 16-state tiles issued (padding of the last tile of a node included)
    for (size_t l = 0; l + 1 < p.level_first.size(); ++l) {
        const int first = p.level_first[l], n_tiles = p.level_first[l + 1] - first;
        if (n_tiles <= 0) continue;
        a.tiles = p.d_tiles + first;
        const int nft = p.level_nft[l];
        slots += (double)n_tiles * nft;
        if (p.level_nrt[l] > 0) {
            // k2c_gemm: 16 * nft states per workgroup, one wave per nrt row tiles
            const int nrt = p.level_nrt[l], RT = (c->C + 15) / 16;
            const int block = 64 * ((RT + nrt - 1) / nrt);
            int gs = (256 * nft + block - 1) / block;
            gs = gs <= 1 ? 1 : (gs <= 2 ? 2 : 4);
            const void* fn = k2c_gemm_kernel(nft, nrt, gs, block > 512 ? 1024 : 512);
            if (!fn) return fail("internal: no k2c_gemm instantiation for nst=%d nrt=%d gs=%d block=%d", nft, nrt, gs, block);
            K2cArgs g = a;
            g.block_threads = block;
            // (a level of one or two rounds of tiles runs 5 % faster in dispatch order: configs[1] 38.5 -> 36.3 us)
            g.xcd_remap = (c->opt.k2c_xcd && n_tiles >= 4 * std::max(c->n_cu, 1)) ? 1 : 0;
            const size_t lds = (size_t)2 * 16 * nft * 34 * sizeof(double);
            if (grant_lds(c, fn, lds, 64 * 1024)) return -1;
            if (launch_kernel(fn, dim3(n_tiles, n_sets), dim3(block), lds, c->stream, g)) return -1;
            continue;
        }
        const bool pair = k2c_pairs(c, (long long)n_tiles * n_sets);
        int nrt_w = 0;
        const int wr = k2c_wave_rows(c, &nrt_w, pair);
        if (launch_k2c_inst(c, k2c_kernel(nft, nrt_w, c->opt.k2c_batch != 0, pair), nft, a, n_tiles, n_sets, 64 * wr)) return -1;
    }
    const double kpad = 4.0 * ((c->C + 3) / 4), rows = 16.0 * ((c->C + 15) / 16);
    c->issued_tables = 2.0 * kpad * rows * 16.0 * slots * n_sets;
    return 0;
}

// ---- MFMA launcher -------------------------------------------------------------------
// Park scratch of a launch (node vectors waiting for their sibling that do not fit LDS): one slot per workgroup that
// can be RESIDENT (occupancy query x CUs, doubled as margin), claimed by the workgroups at run time
// (k2_acquire_park_slot), instead of one region per family tile: at the configs[2] shape 2 x 1,280 slots x 2 parks x
// 33 KB = 169 MB at most instead of 413 MB, and only the slots in use are touched -- they stay in the 256 MB Infinity
// Cache (round 1: 7.7 GB of HBM traffic per launch).  Option k2slots=0 restores one region per tile.
int k2_fit_grid(cafehip_ctx* c, const void* fn, K2MfmaArgs& a, int* grid, int block, size_t lds)
{
    const bool global_parks = walk_sched(c).n_parks > a.lds_parks;
    int slots = 0;
    const bool per_tile = c->opt.k2slots == 0;
    if (global_parks && !per_tile) {
        auto it = c->k2_occ.find({fn, block, lds});
        if (it == c->k2_occ.end()) {
            int nb = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, lds));
            it = c->k2_occ.emplace(std::make_tuple(fn, block, lds), std::max(nb, 1)).first;
        }
        slots = std::min(*grid * a.n_sets, 2 * it->second * std::max(c->n_cu, 1));
    }
    const size_t regions = global_parks ? (size_t)(slots > 0 ? slots : *grid * a.n_sets) : 1;
    const size_t park_bytes = regions * a.n_parks * a.NF * a.LDv * sizeof(double);
    if (park_bytes > c->park_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_park);
        c->d_park = nullptr;
        c->park_cap = 0;
        HIP_TRY(hipMalloc(&c->d_park, park_bytes));
        c->park_cap = park_bytes;
    }
    if (slots > c->park_flags_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_park_flags);
        c->d_park_flags = nullptr;
        HIP_TRY(hipMalloc(&c->d_park_flags, (size_t)slots * sizeof(int32_t)));
        HIP_TRY(hipMemsetAsync(c->d_park_flags, 0, (size_t)slots * sizeof(int32_t), c->stream));   // all free; every owner releases
        c->park_flags_cap = slots;
    }
    a.park = c->d_park;
    a.park_flags = c->d_park_flags;
    a.n_park_slots = slots;
    a.gen_done = nullptr;
    if ((a.col_max != nullptr && c->opt.batch_lockstep > 0) || (a.col_max == nullptr && c->opt.walk_lockstep > 0 && a.n_sets == 1)) {
        // lock-step generations of a batch launch: as many workgroups as the chip holds at once
        auto it = c->k2_occ.find({fn, block, lds});
        if (it == c->k2_occ.end()) {
            int nb = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, lds));
            it = c->k2_occ.emplace(std::make_tuple(fn, block, lds), std::max(nb, 1)).first;
        }
        const int gen = it->second * std::max(c->n_cu, 1);
        if (*grid > gen) {
            if (!c->d_gen_done) HIP_TRY(hipMalloc(&c->d_gen_done, sizeof(int32_t)));
            HIP_TRY(hipMemsetAsync(c->d_gen_done, 0, sizeof(int32_t), c->stream));
            a.gen_done = c->d_gen_done;
            a.gen_size = gen;
            a.gen_slack = (int)((long long)gen * c->opt.batch_lockstep_slack / 100);
        }
    }
    c->k2_grid = *grid;
    c->k2_park_slots = slots;
    return 0;
}

int launch_mfma16(cafehip_ctx* c, K2MfmaArgs a, int nft_w, int nrt_w, int grid, int block, size_t lds)
{
    // only the (NFT_W, NRT_W) pairs within the register budget (NFT_W * NRT_W <= 8 accumulator tiles, NRT_W <= 7:
    // no scratch spills) are instantiated
    const bool objective = c->opt.k2_objective_kernels && a.col_max == nullptr && (a.err == nullptr || a.PTfold != nullptr);
    const void* fn = objective ? k2_mfma16_objective_kernel(nft_w, nrt_w) : k2_mfma16_kernel(nft_w, nrt_w);
    if (!fn) return fail("unsupported 16x16 wave grid NFT_W=%d NRT_W=%d", nft_w, nrt_w);
    if (grant_lds(c, fn, lds, 64 * 1024)) return -1;
    if (k2_fit_grid(c, fn, a, &grid, block, lds)) return -1;
    return launch_kernel(fn, dim3(grid, a.n_sets), dim3(block), lds, c->stream, a);
}

// The park buffers (node vectors waiting for their sibling) stay in LDS when the whole set still lets two
// workgroups share a CU; otherwise they live in global scratch and are fetched back when consumed.
size_t mfma_lds_bytes_with(const cafehip_ctx* c, int nf, int lds_parks)
{
    return (size_t)nf * c->LDv * sizeof(double) * (1 + lds_parks) + k2_scratch_bytes(nf, walk_cols(c), (int)walk_sched(c).ops.size());
}

// Number of park buffers (node vectors waiting for their sibling; slot 0 is the busiest) kept in LDS behind the
// working buffer; the rest live in global scratch and are fetched back when consumed.  LDS parks save the global
// round trip but cost residency: they are used only as far as the CU still holds as many workgroups as the grid
// can put on it (measured at the configs[3] shape: one LDS park at the price of 5 -> 3 workgroups per CU is
// 25 % slower; at configs[1], where 500 workgroups give every CU two either way, it is 4 % faster).
// Option ldspark=<n> overrides (0 = none).
int mfma_lds_parks(const cafehip_ctx* c, int nf, int n_items)
{
    const int n_parks = walk_sched(c).n_parks;
    if (n_parks <= 0) return 0;
    if (c->opt.ldspark >= 0) return std::min(c->opt.ldspark, n_parks);
    const size_t cu_lds = 160 * 1024;
    const int grid = (n_items + nf - 1) / nf;
    const int wanted = std::max(1, (grid + c->n_cu - 1) / std::max(c->n_cu, 1));
    const int resident0 = (int)(cu_lds / std::max<size_t>(mfma_lds_bytes_with(c, nf, 0), 1));
    const int keep = std::max(1, std::min(wanted, resident0));
    int n = 0;
    while (n < n_parks && (int)(cu_lds / mfma_lds_bytes_with(c, nf, n + 1)) >= keep &&
           mfma_lds_bytes_with(c, nf, n + 1) <= (size_t)c->lds_limit)
        ++n;
    return n;
}

size_t mfma_lds_bytes(const cafehip_ctx* c, int nf, int n_items)
{
    return mfma_lds_bytes_with(c, nf, mfma_lds_parks(c, nf, n_items));
}

// Cost model fitted to sweeps on MI355X (tools/sweep_k2.py): every workgroup is resident at once,
// block b lands on CU b % n_cu, its waves go to consecutive SIMDs from a rotating start; the kernel takes
// as long as the busiest SIMD, times a per-wave-count factor (1-2 waves hide less latency, 8 waves pay
// wider barriers), a matrix re-streaming term and an intra-workgroup imbalance term (waves meet at
// barriers: 3,3,2,2 row tiles is 16 % slower than 5,5,5,5 on the cfg4 shape).  `groups` = 4-family groups
// per wave (4 per 16-family tile), so both MFMA shapes are priced in the same unit.
double k2_cost(const cafehip_ctx* c, int n_items, int nf, int groups, int wf, int wr, int RTc)
{
    const int n_cu = std::max(c->n_cu, 1);
    const int W = wf * wr;
    const long n_wg = (n_items + nf - 1) / nf;
    const int wg_on_cu = (int)((n_wg + n_cu - 1) / n_cu);  // busiest CU
    // accumulator-tile steps the busiest CU issues per k-step, spread over its 4 SIMDs (where the waves of
    // several resident workgroups land is not under our control; the intra-workgroup term below prices the
    // uneven deals)
    double per_wg = 0;
    int active_waves = 0;
    for (int w = 0; w < W; ++w) {
        const int wrow = w / wf;
        const int act = RTc / wr + (wrow < RTc % wr ? 1 : 0);  // even deal of the row tiles
        per_wg += act * groups;
        active_waves += act > 0;
    }
    const double simds = std::min(4, std::max(1, wg_on_cu * active_waves));  // a lone 2-wave workgroup uses 2 SIMDs
    const double maxload = wg_on_cu * per_wg / simds;
    // constants re-fitted on tools/sweep_k2*.py data (tools/fit_k2_cost.py) after waves stopped issuing dummy columns
    static const double wpen[9] = {0, 1.1, 1.0, 1.0, 1.0, 0.975, 0.95, 0.925, 0.9};
    double cost = maxload * wpen[W];
    cost *= 1.0 + 0.001 * (n_wg * wf) / (double)n_cu;
    const int hi_t = RTc / wr + (RTc % wr ? 1 : 0);
    const double mean_t = (double)RTc / wr;
    cost *= 1.0 + 0.2 * (hi_t / mean_t - 1.0);
    return cost;
}

// 16x16x4 shape: NF = 16 * nft_w * wf.  Option k2cfg="nftw,nrtw,wf,wr" overrides (tuning sweeps).
bool choose_mfma_cfg(const cafehip_ctx* c, int n_items, K2Cfg* out, double* out_cost, std::vector<K2Cand>* all = nullptr)
{
    const int RT = (std::max(c->C, c->R) + 15) / 16;
    const int RTc = (c->C + 15) / 16;
    if (c->opt.have_cfg16) {
        const K2Cfg k{c->opt.cfg16[0], c->opt.cfg16[1], c->opt.cfg16[2], c->opt.cfg16[3]};
        if (k2_fits16(k.nft_w, k.nrt_w) && k.wf * k.wr >= 1 && k.wf * k.wr <= 8 && k.wr * k.nrt_w >= RT &&
            mfma_lds_bytes(c, 16 * k.nft_w * k.wf, n_items) <= (size_t)c->lds_limit) {
            *out = k;
            *out_cost = 0;
            return true;
        }
    }
    double best = 1e300;
    bool found = false;
    for (int wr = 1; wr <= 8; wr *= 2) {
        const int nrt_w = (RT + wr - 1) / wr;
        if (nrt_w > 7) continue;  // register budget: NFT_W * NRT_W <= 8 accumulator tiles, no spills
        for (int nft_w = 1; nft_w <= 2; ++nft_w) {
            if (nft_w * nrt_w > kMaxTiles16) continue;
            for (int wf = 1; wf * wr <= 8; wf *= 2) {
                const int nf = 16 * nft_w * wf;
                if (mfma_lds_bytes(c, nf, n_items) > (size_t)c->lds_limit) continue;
                const double cost = k2_cost(c, n_items, nf, 4 * nft_w, wf, wr, RTc);
                if (all) all->push_back(K2Cand{cost, false, K2Cfg{nft_w, nrt_w, wf, wr}});
                if (cost < best) {
                    best = cost;
                    *out = K2Cfg{nft_w, nrt_w, wf, wr};
                    found = true;
                }
            }
        }
    }
    *out_cost = best;
    return found;
}

// 4x4x4_4b shape: NF = 4 * G * wf (K2Cfg.nft_w carries G).  Option k2cfg4="G,nrtw,wf,wr" overrides.
bool choose_mfma4_cfg(const cafehip_ctx* c, int n_items, K2Cfg* out, double* out_cost, std::vector<K2Cand>* all = nullptr)
{
    const int RT = (std::max(c->C, c->R) + 15) / 16;
    const int RTc = (c->C + 15) / 16;
    if (c->opt.have_cfg4) {
        const K2Cfg k{c->opt.cfg4[0], c->opt.cfg4[1], c->opt.cfg4[2], c->opt.cfg4[3]};
        if (k2_fits4(k.nft_w, k.nrt_w) && k.wf * k.wr >= 1 && k.wf * k.wr <= 8 &&
            k.wr * k.nrt_w >= RT && mfma_lds_bytes(c, 4 * k.nft_w * k.wf, n_items) <= (size_t)c->lds_limit) {
            *out = k;
            *out_cost = 0;
            return true;
        }
    }
    double best = 1e300;
    bool found = false;
    for (int wr = 1; wr <= 8; wr *= 2) {
        const int nrt_w = (RT + wr - 1) / wr;
        if (nrt_w > 7) continue;
        for (int G = 1; G <= 8; ++G) {
            if (!k2_fits4(G, nrt_w)) continue;
            for (int wf = 1; wf * wr <= 8 && wf <= 2; wf *= 2) {
                const int nf = 4 * G * wf;
                if (mfma_lds_bytes(c, nf, n_items) > (size_t)c->lds_limit) continue;
                // measured: per flop this shape runs ~7 % behind the 16x16x4 one inside the kernel, and
                // few groups per wave amortise the B-operand loads badly (G = 1: 2x, G = 2: 1.2x)
                const double cost = 1.07 * (1.0 + 0.5 / (G * G)) * k2_cost(c, n_items, nf, G, wf, wr, RTc);
                if (all) all->push_back(K2Cand{cost, true, K2Cfg{G, nrt_w, wf, wr}});
                if (cost < best) {
                    best = cost;
                    *out = K2Cfg{G, nrt_w, wf, wr};
                    found = true;
                }
            }
        }
    }
    *out_cost = best;
    return found;
}

int launch_mfma4_g(cafehip_ctx* c, K2MfmaArgs a, int G, int nrt_w, int grid, int block, size_t lds)
{
    // only the (G, NRT_W) pairs within the register budget are instantiated
    // an objective evaluation (no per-row column limits, error model folded into the matrices or absent) runs the instantiation
    // without the batch mode's code (option k2_objective_kernels)
    const bool objective = c->opt.k2_objective_kernels && a.col_max == nullptr && (a.err == nullptr || a.PTfold != nullptr);
    const void* fn = objective ? k2_mfma4_objective_kernel(G, nrt_w) : k2_mfma4_kernel(G, nrt_w);
    // at most 64 root sizes (the reference's test1 table, its example): the instantiation with a lane per family in the posterior
    // epilogue (k2_walk4s.hip; test1 walk 36.1 -> 32.3 us); never in batch mode, whose "epilogue" copies root rows
    if (c->opt.k2_small_r && objective && a.R <= 64 && a.NF <= 96 && !a.skip_epilogue)
        if (const void* fs = k2_mfma4_small_r_kernel(G, nrt_w, 1)) fn = fs;
    if (!fn) return fail("unsupported 4x4 wave grid G=%d NRT_W=%d", G, nrt_w);
    if (grant_lds(c, fn, lds, 64 * 1024)) return -1;
    if (k2_fit_grid(c, fn, a, &grid, block, lds)) return -1;
    return launch_kernel(fn, dim3(grid, a.n_sets), dim3(block), lds, c->stream, a);
}

// measured wave-grid choices of this process, by problem shape
// Round 6: the measurement REPLACES launches instead of adding them.  Rounds 1-4 used to time 4 back-to-back launches of a
// candidate per deciding evaluation (a single ~55 us launch is within the noise of the candidates' differences): ~30 tuning
// evaluations x 3 extra launches made the FIRST search of a process twice as long as the fifth (configs[1]: 13.8 against
// 6.95 ms, profiles/r05_lookahead_ab.txt).  Now every tuning evaluation launches its candidate once -- it costs what the
// candidate is slower than the best grid, ~10-20 us -- and the noise is met with more rounds of the survivors (minimum kept).
constexpr int kTuneReps = 1;
constexpr int kTuneRounds = 8;   // round 0 warm-up, round 1 every grid, rounds 2-7 those within 6 % of the best (minimum kept)
std::mutex g_tuned_mu;
std::map<std::array<long, 8>, K2Cand> g_tuned;
std::array<long, 8> tune_key(const cafehip_ctx* c, int n_items)
{
    return {(long)c->device, (long)n_items, (long)c->C, (long)c->R, (long)walk_cols(c), (long)walk_sched(c).ops.size(),
            (long)walk_sched(c).n_parks, (long)(c->d_err != nullptr)};
}

int launch_k2_mfma(cafehip_ctx* c, const K2Args& v1, int n_items, int n_sets = 1)
{
    if (n_items <= 0) return 0;
    K2Cfg k16{}, k4{};
    double cost16 = 1e300, cost4 = 1e300;
    const bool shape_env = c->opt.mfma != 0;
    const bool allow16 = c->opt.mfma != 4;
    const bool allow4 = c->opt.mfma != 16;
    const bool have16 = allow16 && choose_mfma_cfg(c, n_items, &k16, &cost16);
    const bool have4 = allow4 && choose_mfma4_cfg(c, n_items, &k4, &cost4);
    if (!have16 && !have4) {
        // matrices too large for the MFMA wave grids: the row-per-thread kernel handles them
        if (n_sets > 1) return fail("several parameter sets per pass need the matrix-core kernel (matrix side too large)");
        if (c->walk_compressed) return fail("internal: compressed walk without a matrix-core wave grid");
        c->k2_used_mfma = false;
        K2Args a1 = v1;
        return launch_k2_v1(c, a1, n_items);
    }
    bool use4 = have4 && (!have16 || cost4 < cost16);
    K2Cfg k = use4 ? k4 : k16;
    // The cost model ranks the wave grids to within ~10 %; every grid produces bit-identical values (same
    // accumulation order), so the objective path MEASURES its few best candidates on the first evaluations of a
    // table (each of them a normal, valid evaluation) and keeps the fastest.  option k2tune=0 disables;
    // explicit k2cfg / k2cfg4 / mfma options do too.
    bool tuning_launch = false;
    {
        const bool overridden = !c->opt.k2tune || shape_env || c->opt.have_cfg16 || c->opt.have_cfg4;
        const bool enabled = n_sets == 1 && v1.col_max == nullptr && !overridden;
        auto& t = c->tune;
        if (!enabled && n_sets > 1) {
            // several parameter sets in one pass: no measurement; the grid a single-set evaluation settled on is
            // kept if there is one, else the cost model's choice stands
            if (t.n_items == n_items && t.locked >= 0 && !t.cands.empty()) {
                use4 = t.cands[t.locked].use4;
                k = t.cands[t.locked].cfg;
            }
        } else if (!enabled && v1.col_max != nullptr && !overridden) {
            // batch mode (Monte-Carlo null rows: same tree, same matrices, another row count): one launch cannot be
            // measured against alternatives; the grid the table's evaluations settled on beats the model's guess
            // (cfg 5 null, 250 k rows: 21.1 ms with the model's 2,2,1,8, 16.5 ms with the table's 1,4,2,4)
            auto fits = [&](const K2Cand& cd) {
                const int nf_c = cd.use4 ? 4 * cd.cfg.nft_w * cd.cfg.wf : 16 * cd.cfg.nft_w * cd.cfg.wf;
                return mfma_lds_bytes(c, nf_c, n_items) <= (size_t)c->lds_limit;   // (the table's walk may have been a reduced one)
            };
            if (t.locked >= 0 && !t.cands.empty() && fits(t.cands[t.locked])) {
                use4 = t.cands[t.locked].use4;
                k = t.cands[t.locked].cfg;
            } else {
                // ... or on for another table of this shape earlier in the process (nearest row count)
                std::lock_guard<std::mutex> g(g_tuned_mu);
                const auto want = tune_key(c, n_items);
                double best_d = 1e300;
                for (const auto& kv : g_tuned) {
                    bool same = kv.first[0] == want[0];
                    for (int i = 2; i < 8; ++i) same = same && kv.first[i] == want[i];
                    if (!same || !fits(kv.second)) continue;
                    const double d = fabs(log((double)std::max(kv.first[1], 1L) / (double)n_items));
                    if (d < best_d) {
                        best_d = d;
                        use4 = kv.second.use4;
                        k = kv.second.cfg;
                    }
                }
            }
            // A large batch runs better with two family groups of waves per workgroup: the second group shares the
            // matrix operand through the CU's L1 and the trimmed tiles keep twice the waves busy (cfg 5 null on the
            // table's 1,4,1,4: 11.7 ms, on 1,4,2,4: 11.1 ms; profiles/r03/mcnull_trimmed_counts_grids_mixing.txt)
            if (!use4 && k.wf == 1 && 2 * k.wr <= 8 && n_items >= 8L * 32 * k.nft_w * std::max(c->n_cu, 1) &&
                mfma_lds_bytes(c, 32 * k.nft_w, n_items) <= (size_t)c->lds_limit)
                k.wf = 2;
        } else if (!enabled) {
            t.n_items = -1;
        } else {
            if (t.n_items != n_items) {  // new table (set_families / set_tree reset n_items to -1)
                t.n_items = n_items;
                t.cands.clear();
                std::vector<K2Cand> all16, all4;
                K2Cfg dummy;
                double dc;
                choose_mfma_cfg(c, n_items, &dummy, &dc, &all16);
                choose_mfma4_cfg(c, n_items, &dummy, &dc, &all4);
                auto by_cost = [](const K2Cand& x, const K2Cand& y) { return x.cost < y.cost; };
                std::sort(all16.begin(), all16.end(), by_cost);
                std::sort(all4.begin(), all4.end(), by_cost);
                // five per shape, but none the model itself prices more than 35 % above its best
                double floor_cost = 1e300;
                if (!all16.empty()) floor_cost = std::min(floor_cost, all16[0].cost);
                if (!all4.empty()) floor_cost = std::min(floor_cost, all4[0].cost);
                for (size_t i = 0; i < all16.size() && i < 5; ++i)
                    if (i == 0 || all16[i].cost <= 1.35 * floor_cost) t.cands.push_back(all16[i]);
                for (size_t i = 0; i < all4.size() && i < 5; ++i)
                    if (i == 0 || all4[i].cost <= 1.35 * floor_cost) t.cands.push_back(all4[i]);
                t.best_ms.assign(t.cands.size(), 1e30f);
                t.cur = t.round = 0;
                t.locked = t.cands.size() <= 1 ? 0 : -1;
                {  // a table of the same shape was measured before in this process (e.g. lhtest's simulated tables)
                    std::lock_guard<std::mutex> g(g_tuned_mu);
                    auto it = g_tuned.find(tune_key(c, n_items));
                    if (it != g_tuned.end()) {
                        t.cands.assign(1, it->second);
                        t.best_ms.assign(1, 0.0f);
                        t.locked = 0;
                    }
                }
                t.pending = false;
                if (!t.e0) {
                    HIP_TRY(hipEventCreate(&t.e0));
                    HIP_TRY(hipEventCreate(&t.e1));
                }
            }
            if (t.locked < 0 && t.pending) {  // collect the previous evaluation's measurement
                HIP_TRY(hipEventSynchronize(t.e1));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, t.e0, t.e1));
                ms /= (float)std::max(t.reps_launched, 1);
                // round 0 runs while the clocks are still ramping up (a grid measured 0.236 ms there and 0.170 ms
                // in steady state): it is a warm-up and eliminates nothing.  Round 1 times every grid (up to kTuneReps
                // launches each, see below), rounds 2-4 once more each those within 5 % of the best so far.
                if (t.round <= 1) t.best_ms[t.cur] = ms;
                else t.best_ms[t.cur] = std::min(t.best_ms[t.cur], ms);
                t.pending = false;
                float best = 1e30f;
                if (t.round >= 1)
                    for (size_t i = 0; i < t.best_ms.size(); ++i)
                        if (t.round >= 2 || (int)i <= t.cur) best = std::min(best, t.best_ms[i]);
                do {
                    if (++t.cur == (int)t.cands.size()) {
                        t.cur = 0;
                        ++t.round;
                    }
                } while (t.round >= 2 && t.round < kTuneRounds && t.best_ms[t.cur] > 1.06f * best);
                if (t.round >= kTuneRounds) {
                    t.locked = (int)(std::min_element(t.best_ms.begin(), t.best_ms.end()) - t.best_ms.begin());
                    if (c->opt.k2tune_log)
                        for (size_t i = 0; i < t.cands.size(); ++i)
                            fprintf(stderr, "cafehip: wave grid %s %d,%d,%d,%d  model %.3g  measured %.4f ms%s\n", t.cands[i].use4 ? "4x4" : "16x16",
                                    t.cands[i].cfg.nft_w, t.cands[i].cfg.nrt_w, t.cands[i].cfg.wf, t.cands[i].cfg.wr, t.cands[i].cost, t.best_ms[i],
                                    (int)i == t.locked ? "  <- kept" : "");
                    std::lock_guard<std::mutex> g(g_tuned_mu);
                    g_tuned[tune_key(c, n_items)] = t.cands[t.locked];
                }
            }
            if (!t.cands.empty()) {
                const K2Cand& pick = t.cands[t.locked >= 0 ? t.locked : t.cur];
                use4 = pick.use4;
                k = pick.cfg;
                tuning_launch = t.locked < 0;
            }
        }
    }
    const int nf = use4 ? 4 * k.nft_w * k.wf : 16 * k.nft_w * k.wf;
    const int grid = (n_items + nf - 1) / nf;
    const int block = 64 * k.wf * k.wr;
    const size_t lds = mfma_lds_bytes(c, nf, n_items);
    K2MfmaArgs a;
    memset(&a, 0, sizeof a);
    a.PT = v1.PT;
    a.node_key = v1.node_key;
    a.n_nodes = v1.n_nodes;
    a.prior = v1.prior;
    a.logprior = v1.logprior;
    a.ops = c->walk_compressed ? c->cp.d_ops : c->d_mops;
    a.n_ops = (int)walk_sched(c).ops.size();
    a.n_sets = n_sets;
    a.counts = c->walk_compressed ? c->cp.d_counts : v1.counts;
    a.Fu = v1.Fu;
    a.n_leaves = walk_cols(c);
    if (c->walk_compressed) {
        a.tables = c->cp.d_tables;
        a.table_off = c->cp.d_table_off;
        a.table_set_stride = c->cp.table_elems;
    }
    a.C = v1.C;
    a.R = v1.R;
    a.root_min = v1.root_min;
    a.LD = v1.LD;
    a.KP = v1.KP;
    a.LDv = v1.LDv;
    a.ksteps = (c->C + 3) / 4;
    a.Wf = k.wf;
    a.Wr = k.wr;
    a.NF = nf;
    a.park = nullptr;   // sized and set by k2_fit_grid for the grid actually launched
    a.n_parks = std::max(walk_sched(c).n_parks, 1);
    a.lds_parks = mfma_lds_parks(c, nf, n_items);
    a.err = v1.err;
    a.err_ld = v1.err_ld;
    a.leaf_has_err = (c->walk_compressed && v1.leaf_has_err) ? c->cp.d_col_has_err : v1.leaf_has_err;
    a.err_banded = (v1.err != nullptr) ? c->err_banded : 0;
    a.err_dlo = c->err_dlo;
    a.err_dhi = c->err_dhi;
    a.PTfold = (v1.err != nullptr && v1.col_max == nullptr && c->fold_current) ? c->d_PTfold : nullptr;
    a.root_lo = v1.root_lo;
    a.root_hi = v1.root_hi;
    a.col_max = v1.col_max;
    a.out_off = v1.out_off;
    a.out_root = v1.out_root;
    a.trim = (v1.col_max != nullptr && c->opt.batch_trim) ? 1 : 0;
    a.max_lik = v1.max_lik;
    a.argmax = v1.argmax;
    a.max_post = v1.max_post;
    a.skip_epilogue = (c->opt.k2_skip_epilogue && v1.col_max == nullptr) ? 1 : 0;
    c->k2_cfg[0] = k.nft_w;
    c->k2_cfg[1] = k.nrt_w;
    c->k2_cfg[2] = k.wf;
    c->k2_cfg[3] = k.wr;
    c->k2_nf = nf;
    c->k2_block = block;
    c->k2_lds = lds;
    c->k2_used_mfma = true;
    c->k2_shape4 = use4;
    {
        // matrix-instruction flops this launch issues: one product per internal child, roundup16(rows) x roundup4(C)
        // per family slot (tile padding included)
        const double kpad = 4.0 * a.ksteps;
        double per_slot = 0;
        for (const auto& op : walk_sched(c).ops) {
            const double rows = 16.0 * (((op.is_root ? c->R : c->C) + 15) / 16);
            per_slot += 2.0 * kpad * rows * ((op.kind[0] == 1) + (op.kind[1] == 1));
        }
        c->issued_walk = per_slot * (double)nf * grid * n_sets;
    }
#ifdef CAFE_K2_STAMPS
    const char* stamps_file = getenv("CAFEHIP_STAMPS_FILE");
    const size_t stamps_n = (size_t)grid * 8 * K2_STAMP_SLOTS;
    if (stamps_file) {
        if (stamps_n > c->stamps_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            hipFree(c->d_stamps);
            c->d_stamps = nullptr;
            HIP_TRY(hipMalloc(&c->d_stamps, stamps_n * sizeof(unsigned long long)));
            c->stamps_cap = stamps_n;
        }
        HIP_TRY(hipMemsetAsync(c->d_stamps, 0, stamps_n * sizeof(unsigned long long), c->stream));
        a.stamps = c->d_stamps;
    }
#endif
    if (tuning_launch) HIP_TRY(hipEventRecord(c->tune.e0, c->stream));
    // a deciding measurement (rounds 1, 2) of a SHORT launch times several back-to-back launches: the walk is
    // idempotent, and one launch of a small table (~0.1 ms) is within the noise of the candidates' differences.  A
    // launch of a millisecond is its own measurement (the extra launches of ten candidates would cost a one-off
    // search of ~150 evaluations 30 % of its time)
    int reps = 1;
    if (tuning_launch && c->tune.round >= 1) {
        const float warm = c->tune.best_ms[c->tune.cur];   // round 0's (or round 1's) time of this grid
        reps = warm < 0.25f ? kTuneReps : 1;
        (void)warm;
    }
    if (tuning_launch) c->tune.reps_launched = reps;
    {
        // (probe: CAFEHIP_K2_REPS=2 launches the idempotent walk twice back to back, so that a kernel trace shows what a launch
        // costs whose code is already in the instruction caches: profiles/r06/walk_warm_icache_probe.txt)
        static const int extra_reps = [] { const char* e = getenv("CAFEHIP_K2_REPS"); return e ? atoi(e) : 0; }();
        if (!tuning_launch && extra_reps > 1) reps = extra_reps;
    }
    int rc = 0;
    for (int rep = 0; rep < reps && rc == 0; ++rep) {
        if (use4) rc = launch_mfma4_g(c, a, k.nft_w, k.nrt_w, grid, block, lds);
        else rc = launch_mfma16(c, a, k.nft_w, k.nrt_w, grid, block, lds);
    }
    if (rc == 0 && tuning_launch) {
        HIP_TRY(hipEventRecord(c->tune.e1, c->stream));
        c->tune.pending = true;
    }
#ifdef CAFE_K2_STAMPS
    if (rc == 0 && stamps_file) {
        // header: grid, waves per workgroup, slots, n_ops, NF, shape (4 / 16), then the raw stamps (overwritten per launch)
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<unsigned long long> h(stamps_n);
        HIP_TRY(hipMemcpy(h.data(), c->d_stamps, stamps_n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(stamps_file, "wb")) {
            const long long hdr[8] = {grid, block / 64, K2_STAMP_SLOTS, a.n_ops, nf, use4 ? 4 : 16, k.wf, k.wr};
            fwrite(hdr, sizeof hdr, 1, f);
            fwrite(walk_sched(c).ops.data(), sizeof(cafehip::MfmaOp), walk_sched(c).ops.size(), f);
            fwrite(h.data(), sizeof(unsigned long long), stamps_n, f);
            fclose(f);
        }
    }
#endif
    return rc;
}

int launch_k2(cafehip_ctx* c, K2Args& a, int n_items, int n_sets = 1)
{
    if (c->opt.k2 != 0) {
        c->k2_used_mfma = false;
        if (n_sets > 1) return fail("several parameter sets per pass need the matrix-core kernel");
        return launch_k2_v1(c, a, n_items);
    }
    return launch_k2_mfma(c, a, n_items, n_sets);
}

void fill_common_k2(cafehip_ctx* c, K2Args& a)
{
    memset(&a, 0, sizeof a);
    a.PT = c->d_PT;
    a.node_key = c->cur_node_key;
    a.n_nodes = c->n_nodes;
    a.prior = c->d_prior;
    a.logprior = c->d_logprior;
    a.ops = c->d_ops;
    a.n_ops = (int)c->sched.ops.size();
    a.n_leaves = c->n_leaves;
    a.C = c->C;
    a.R = c->R;
    a.root_min = c->root_min;
    a.LD = c->LD;
    a.KP = c->KP;
    a.LDv = c->LDv;
}

