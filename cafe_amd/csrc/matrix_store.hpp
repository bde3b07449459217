// matrix_store.hpp -- everything about the transition matrices on the device EXCEPT the kernel that computes them:
// their storage (demand region + the entries built ahead of time), the staging of an evaluation's keys into the pinned
// parameter ring, the K1 / error-fold launches, and the store of parameter sets built ahead (context.hpp, MatrixCache;
// cafehip_prefetch_matrices).  Part of cafehip.hip (included inside its anonymous namespace; round-5 split).
#pragma once
// ---- matrices of sets that may be evaluated next: bookkeeping (context.hpp, MatrixCache) ------------------------------
void mc_invalidate(cafehip_ctx* c)
{
    // (whatever the entries were built for -- tree, ranges, error model, arithmetic form -- has changed, or their slots moved)
    for (auto& e : c->mc.e) e.valid = false;
    if (c->mc.bound >= 0) c->have_matrices = false;   // the bound matrices were an entry's
    c->mc.bound = -1;
    c->mc.pending_sets = 0;
    c->cur_node_key = c->d_node_key;
}

// both streams idle (before storage the speculative builds write is released or moved)
int sync_streams(cafehip_ctx* c)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->mc.stream) HIP_TRY(hipStreamSynchronize(c->mc.stream));
    return 0;
}

size_t mc_slots(const cafehip_ctx* c) { return c->mc.e.size() * (size_t)c->mc.kpe; }

// d_PT = [demand region: pt_keys_cap slots][cache entries: mc_slots(c) slots], one matrix [KP][LD] per slot
int ensure_matrix_storage(cafehip_ctx* c, size_t min_keys = 0)
{
    const size_t need_keys = std::max((size_t)std::max(c->n_nodes, 1), min_keys);
    if (c->d_PT && c->pt_keys_cap >= need_keys && c->mc.slots_allocated == mc_slots(c)) return 0;
    if (sync_streams(c)) return -1;
    mc_invalidate(c);
    // the buffer moves and is zero-filled: whatever was built on demand is gone too (a later cafehip_get_matrix, root-likelihood
    // or Viterbi call must rebuild instead of reading zeros -- ADVICE r05)
    c->have_matrices = false;
    c->fold_current = false;
    hipFree(c->d_PT);
    c->d_PT = nullptr;
    const size_t keep = std::max(need_keys, c->pt_keys_cap);
    // (+ 16 rows behind the last slot: k2c_gemm's operand ring requests up to three k-steps past a matrix's last and never uses them)
    const size_t bytes = ((keep + mc_slots(c)) * (size_t)c->KP + 16) * c->LD * sizeof(double);
    HIP_TRY(hipMalloc(&c->d_PT, bytes));
    // padding rows/cols stay zero forever; ordered on the context's (non-blocking) stream, where K1 will run
    HIP_TRY(hipMemsetAsync(c->d_PT, 0, bytes, c->stream));
    if (!c->mc.e.empty()) HIP_TRY(hipStreamSynchronize(c->stream));   // (the speculation stream is not ordered behind that fill)
    c->pt_keys_cap = keep;
    c->mc.first_slot = keep;
    c->mc.slots_allocated = mc_slots(c);
    return 0;
}

// the error-folded twins live at the same slots of d_PTfold
int ensure_fold_storage(cafehip_ctx* c)
{
    const size_t need = (c->pt_keys_cap + c->mc.slots_allocated) * (size_t)c->KP * c->LD * sizeof(double);
    if (c->d_PTfold && c->ptfold_cap == need) return 0;
    if (sync_streams(c)) return -1;
    for (auto& e : c->mc.e) e.folded = false;
    hipFree(c->d_PTfold);
    c->d_PTfold = nullptr;
    c->ptfold_cap = 0;
    c->fold_current = false;
    HIP_TRY(hipMalloc(&c->d_PTfold, need));
    HIP_TRY(hipMemsetAsync(c->d_PTfold, 0, need, c->stream));  // rows beyond C stay zero
    if (!c->mc.e.empty()) HIP_TRY(hipStreamSynchronize(c->stream));
    c->ptfold_cap = need;
    return 0;
}

// k1e_fold_error over `nkeys` matrices from slot `first` on: PTfold[slot] = error model folded into PT[slot]
int launch_fold_slots(cafehip_ctx* c, hipStream_t stream, size_t first, int nkeys)
{
    const size_t off = first * (size_t)c->KP * c->LD;
    dim3 grid((c->LD + 255) / 256, c->C, nkeys);
    FoldArgs fa{c->d_PT + off, c->d_PTfold + off, c->d_err, c->err_mfs + 1, c->err_banded, c->err_dlo, c->err_dhi, c->C, c->KP, c->LD};
    return launch_kernel(k1e_fold_kernel(), grid, dim3(256), 0, stream, fa);
}

// Posterior mode with an error model: fold it into this evaluation's matrices (k1e_fold_error), so that every
// leaf stays a column gather.  Option errfold=0 keeps the per-family sums (A/B runs).
int launch_error_fold(cafehip_ctx* c)
{
    c->fold_current = false;
    if (!c->d_err || c->nkeys == 0) return 0;
    if (!c->opt.errfold) return 0;
    if (ensure_fold_storage(c)) return -1;
    if (launch_fold_slots(c, c->stream, 0, c->nkeys)) return -1;
    c->fold_current = true;
    return 0;
}

// node -> matrix maps on the device: rows [0, kMaxSets) for the sets of an evaluation built on demand, row kMaxSets + e for
// cache entry e (written by the K1 launch that builds the entry)
int ensure_node_key_store(cafehip_ctx* c)
{
    const int rows = kMaxSets + (int)c->mc.e.size();
    if (c->d_node_key && c->node_key_rows == rows) return 0;
    if (sync_streams(c)) return -1;
    mc_invalidate(c);
    hipFree(c->d_node_key);
    c->d_node_key = nullptr;
    const size_t bytes = (size_t)rows * c->n_nodes * sizeof(int32_t);
    HIP_TRY(hipMalloc(&c->d_node_key, bytes));
    // (ordered on the context's stream, where K1 will write the map: a null-stream memset is not ordered with a
    // non-blocking stream and could land AFTER the first evaluation's K1)
    HIP_TRY(hipMemsetAsync(c->d_node_key, 0, bytes, c->stream));
    if (!c->mc.e.empty()) HIP_TRY(hipStreamSynchronize(c->stream));
    c->node_key_rows = rows;
    c->cur_node_key = c->d_node_key;
    return 0;
}

// the parameter ring is sized by the tree: (re)allocated by cafehip_set_tree
int ensure_param_ring(cafehip_ctx* c)
{
    const int key_cap = kMaxSets * std::max(c->n_nodes - 1, 1);
    const size_t bytes = eval_block_bytes(key_cap, c->n_nodes);
    if (c->h_params[0] && bytes <= c->ring_bytes && key_cap == c->key_cap) return 0;
    if (sync_streams(c)) return -1;
    for (int i = 0; i < kParamRing; ++i) {
        if (c->h_params[i]) hipHostFree(c->h_params[i]);
        c->h_params[i] = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->h_params[i], bytes, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->h_params[i], 0, bytes);
    }
    c->key_cap = key_cap;
    c->ring_bytes = bytes;
    hipFree(c->d_node_key);
    c->d_node_key = nullptr;
    c->node_key_rows = 0;
    return ensure_node_key_store(c);
}

// one (int branch length, lambda, mu) key reduced to the scalars K1 needs; `slot`: where its matrix goes in d_PT
void fill_key(const cafehip_ctx* c, KeyParam& key, int bl, double lambda, double mu, int slot)
{
    const cafehip::KeyScalars ks = cafehip::key_scalars(bl, lambda, mu);
    key.log_alpha = ks.log_alpha;
    key.log_beta = ks.log_beta;
    key.log_coeff = ks.log_coeff;
    key.coeff = ks.coeff;
    key.mode = ks.mode;
    key.bl = bl;
    key.l2a = ks.l2a;
    key.l2b = ks.l2b;
    key.rho_m = ks.rho_m;
    key.rho_e = ks.rho_e;
    // 2: the register-blocked kernel may run rho^j in plain doubles over 8-term chunks without leaving
    // the double range (binomial products * rho^8 stay below 2^1000); 1: per-term mantissa/exponent form
    key.fast_ok = !ks.fast_ok ? 0 : ((8.0 * std::abs(ks.rho_e) + c->lnc.log2_max_prod + 8.0 < 1000.0) ? 2 : 1);
    key.slot = slot;
}

// host part of reset_birthdeath_cache: unique keys over non-root nodes
// (cafe/cafe_tree.c:374-391, 461-483) -> staged parameter block
int stage_params(cafehip_ctx* c, const double* node_lambda, const double* node_mu,
                 const double* prior, int n_sets = 1, int forced_slot = -1)
{
    if (c->n_nodes <= 0) return fail("no tree set");
    if (c->M < 0) return fail("no families/ranges set");
    if (n_sets < 1 || n_sets > kMaxSets) return fail("1..%d parameter sets per evaluation, got %d", kMaxSets, n_sets);
    // forced_slot: the block a pre-armed chain will read (arm_next reserved it; its event sits BEHIND that chain and must
    // not be waited for here -- the chain waits for us)
    const int slot = forced_slot >= 0 ? forced_slot : c->ring_pos;
    if (forced_slot < 0) {
        c->ring_pos = (c->ring_pos + 1) % kParamRing;
        HIP_TRY(hipEventSynchronize(c->h_params_ev[slot]));
    }
    EvalHeader* h = c->h_params[slot];
    KeyParam* keys = eval_keys(h);
    int32_t* node_key = eval_node_key(h, c->key_cap);
    c->node_key.assign(c->n_nodes, -1);
    int nk = 0;
    auto& kl = c->stage_l;
    auto& km = c->stage_m;
    auto& kb = c->stage_b;
    kl.clear();
    km.clear();
    kb.clear();
    for (int set = 0; set < n_sets; ++set) {
        const double* nl = node_lambda + (size_t)set * c->n_nodes;
        const double* nm = node_mu + (size_t)set * c->n_nodes;
        for (int i = 0; i < c->n_nodes; ++i) {
            node_key[(size_t)set * c->n_nodes + i] = 0;
            if (i == c->root) continue;
            if (!(c->bl[i] > 0))
                return fail("node %d has branch length %g <= 0: the reference binds no matrix to it "
                            "(cafe/cafe_tree.c:341-342)", i, c->bl[i]);
            const int bl = c->bl_int[i];
            int k = 0;
            for (; k < nk; ++k)
                if (kb[k] == bl && kl[k] == nl[i] && km[k] == nm[i]) break;
            if (k == nk) {
                if (nk == c->key_cap) return fail("more than %d distinct matrices in one evaluation", c->key_cap);
                kb.push_back(bl);
                kl.push_back(nl[i]);
                km.push_back(nm[i]);
                fill_key(c, keys[k], bl, nl[i], nm[i], k);
                ++nk;
            }
            if (set == 0) c->node_key[i] = k;
            node_key[(size_t)set * c->n_nodes + i] = k;
        }
    }
    h->nkeys = nk;
    h->n_sets = n_sets;
    h->n_nodes = c->n_nodes;
    h->key_cap = c->key_cap;
    c->nkeys = nk;
    c->all_keys_fast = true;
    for (int k = 0; k < nk; ++k)
        if (keys[k].mode >= 2 && !keys[k].fast_ok) c->all_keys_fast = false;
    if (prior) {
        // compute_posterior adds log(prior[j]) (cafe/lambda.cpp:681); the log is taken on the host -- once per prior:
        // a search hands over the same prior at every evaluation: the device copy is refreshed (by K1, from this block)
        // only in an evaluation whose prior differs from the one on the device
        c->cur_prior_n = 0;
        if (!c->prior_on_device || (int)c->prior_seen.size() != c->R || memcmp(c->prior_seen.data(), prior, sizeof(double) * c->R) != 0) {
            c->prior_seen.assign(prior, prior + c->R);
            c->logprior_seen.resize(c->R);
            for (int j = 0; j < c->R; ++j) c->logprior_seen[j] = std::log(prior[j]);
            double* hp = const_cast<double*>(eval_prior(h, eval_prior_offset(c->key_cap, c->n_nodes)));
            memcpy(hp, c->prior_seen.data(), sizeof(double) * c->R);
            memcpy(hp + kMaxPrior, c->logprior_seen.data(), sizeof(double) * c->R);
            c->cur_prior_n = c->R;
        }
    } else {
        c->cur_prior_n = 0;
    }
    if (ensure_matrix_storage(c, (size_t)nk)) return -1;
    c->cur_params = h;
    c->cur_slot = slot;
    c->cur_sets = n_sets;
    c->mc.bound = -1;                   // the pruning launches read the demand region again
    c->cur_node_key = c->d_node_key;
    return 0;
}

// The pinned block may be rewritten once K1 has consumed it.  Every path that staged a block records the slot's
// event -- on success behind the evaluation's LAST launch (a marker packet between K1 and the next kernel cost ~5 us
// of every evaluation), and on EVERY early return too (a slot left unrecorded would look free to hipEventSynchronize
// eight stagings later while K1 might still be reading it).
struct RingGuard {
    cafehip_ctx* c;
    bool armed = false;
    explicit RingGuard(cafehip_ctx* ctx) : c(ctx) {}
    void arm() { armed = true; }
    int record_now()
    {
        armed = false;
        HIP_TRY(hipEventRecord(c->h_params_ev[c->cur_slot], c->stream));
        return 0;
    }
    ~RingGuard()
    {
        if (armed) (void)hipEventRecord(c->h_params_ev[c->cur_slot], c->stream);
    }
};

// which build of exp() this host's libm runs (exp_like_host.hpp): decided once per process, at the first context's creation
// (200,000 calls of std::exp: a few milliseconds that do not belong inside an evaluation)
int host_exp_variant_once()
{
    static const int variant = host_exp_variant();
    return variant;
}

// One K1 launch: the matrices of the staged block `ep` (nkeys keys, each stored at its own slot of d_PT) on `stream`; block
// (0,0,0) mirrors the node -> matrix map of set s into row set_row[s] of the device store, resets `first_zero` (or NULL) and
// mirrors the prior when n_prior > 0.  `all_fast`: every key qualifies for the product forms.
struct K1Launch {
    hipStream_t stream;
    const EvalHeader* ep;
    int nkeys, n_sets;
    int set_row[kMaxSets];
    int32_t* first_zero;
    int n_prior;
    bool all_fast;
    int kpb = 0;   // keys per workgroup (0: option k1kpb)
};

// the arithmetic form K1 runs for a block of keys (the same for a set built on demand and one built ahead of time)
bool k1_product_form(const cafehip_ctx* c, bool all_fast) { return c->lnc.product_form_ok && all_fast && !c->force_exact && c->opt.k1 != 1; }

// the kernel, grid and arguments of a K1 launch
struct K1Plan {
    const void* fn = nullptr;
    dim3 grid;
    size_t lds = 0;
    bool register_blocked = false;
    K1Args a;
};

int plan_k1_block(cafehip_ctx* c, const K1Launch& L, K1Plan& P)
{
    // table rows are staged once per workgroup and reused for keys_per_block keys; keep >= ~3 workgroups
    // per CU in flight.  Measured: more keys per block only lengthens the heavy tiles (tools/sweep_k1.py)
    const int kpb = L.kpb > 0 ? L.kpb : std::max(1, c->opt.k1_kpb);
    K1Args& a = P.a;
    memset(&a, 0, sizeof a);
    a.ep = L.ep;   // pinned host block
    a.ld_lnc = c->lnc.ld;
    a.PT = c->d_PT;
    a.M = c->M;
    a.LD = c->LD;
    a.KP = c->KP;
    a.first_zero = L.first_zero;
    a.keys_per_block = kpb;
    a.node_key_dev = c->d_node_key;
    for (int q = 0; q < kMaxSets; ++q) a.set_row[q] = L.set_row[q];
    a.n_nodes = c->n_nodes;
    a.n_sets = L.n_sets;
    a.nkeys = L.nkeys;
    a.key_cap = c->key_cap;
    a.n_prior = L.n_prior;
    a.prior_offset = eval_prior_offset(c->key_cap, c->n_nodes);
    a.prior_dev = c->d_prior;
    a.logprior_dev = c->d_logprior;
    a.exp_variant = c->opt.exp_like_host ? host_exp_variant_once() : 0;
    size_t lds = 2 * 16 * (size_t)c->lnc.ld * sizeof(double);
    const bool use_lds = lds <= 150 * 1024;  // bigger tables are read through L1/L2 instead
    if (!use_lds) lds = 0;
    const bool product = k1_product_form(c, L.all_fast);
    const bool blocked = product && c->opt.k1 != 2;
    const int K1Q = k1_rb_columns();
    const size_t lds_rb = 16 * (size_t)((c->lnc.ld + 8) + (c->lnc.ld + k1_rb_bpad() + 8)) * sizeof(double);
    if (blocked && lds_rb <= 150 * 1024) {
        P.fn = k1_rb_kernel();
        P.register_blocked = true;
        P.lds = lds_rb;
        a.tabA = c->d_expA;
        a.tabB = c->d_expB;
        a.gx = (c->S + 16 * K1Q - 1) / (16 * K1Q);
        a.gy = (c->S + 15) / 16;
        a.gz = (L.nkeys + kpb - 1) / kpb;
        a.balanced = 0;
        // (measured: -15 % at 1,280 workgroups -- 62 keys of a 251-wide matrix --, +8 % at 500, where every workgroup is resident
        // from the start and the order only delays the staging of the heavy tiles: profiles/r06/k1_balance_and_epilogue_ablation.txt)
        if (c->opt.k1_balance && a.gx * a.gy <= (int)sizeof a.tile_of_rank && (c->opt.k1_balance > 1 || a.gx * a.gy * a.gz >= 3 * std::max(c->n_cu, 1))) {
            // tiles ranked by work: a thread (row s, columns cb .. cb + K1Q - 1) runs floor(min(s, cb + K1Q - 1, M) / 8) + 1 chunks
            std::vector<std::pair<long, int>> w;
            for (int by = 0; by < a.gy; ++by)
                for (int bx = 0; bx < a.gx; ++bx) {
                    long sum = 0;
                    for (int s = 16 * by; s < 16 * by + 16 && s <= c->M; ++s)
                        for (int cb = 16 * K1Q * bx; cb < 16 * K1Q * (bx + 1) && cb <= c->M; cb += K1Q)
                            sum += s == 0 ? 1 : std::min(s, std::min(cb + K1Q - 1, c->M)) / 8 + 1;
                    w.push_back({-sum, by * a.gx + bx});
                }
            std::sort(w.begin(), w.end());
            for (size_t r = 0; r < w.size(); ++r) a.tile_of_rank[r] = (unsigned char)w[r].second;
            // (1 -- a heavy half, then a light half ascending -- and 2 -- alternating -- measured behind 3, heaviest first throughout,
            // at 62 keys of a 151-wide matrix: 34.7 us in grid order, 34.1 / 33.9 / 30.9 us; profiles/r06/k1_tile_order_ab.txt)
            a.balanced = c->opt.k1_balance >= 10 ? c->opt.k1_balance - 10 : (c->opt.k1_balance > 1 ? c->opt.k1_balance : 3);
        }
        P.grid = dim3(a.gx * a.gy * a.gz);
    } else {
        // product form: every key of this block qualifies and the staged tables are exp(ln C); else the exact form
        P.fn = k1_kernel(use_lds, product);
        P.lds = lds;
        a.tabA = product ? c->d_expA : c->d_lncA;
        a.tabB = product ? c->d_expB : c->d_lncB;
        P.grid = dim3((c->S + 15) / 16, (c->S + 15) / 16, (L.nkeys + kpb - 1) / kpb);
    }
    return 0;
}

int launch_k1_block(cafehip_ctx* c, const K1Launch& L)
{
    if (L.nkeys == 0) return 0;
    K1Plan P;
    if (plan_k1_block(c, L, P)) return -1;
    if (grant_lds(c, P.fn, P.lds, 48 * 1024)) return -1;
    return launch_kernel(P.fn, P.grid, dim3(256), P.lds, L.stream, P.a);
}

// K1 of the evaluation staged by stage_params, on the context's stream, into the demand region
int launch_k1(cafehip_ctx* c, int32_t* d_first_zero = nullptr, bool defer_ring_event = false)
{
    if (c->nkeys == 0) return 0;
    K1Launch L;
    L.stream = c->stream;
    L.ep = c->cur_params;
    L.nkeys = c->nkeys;
    L.n_sets = c->cur_sets;
    for (int q = 0; q < kMaxSets; ++q) L.set_row[q] = q;
    L.first_zero = d_first_zero;
    L.n_prior = c->cur_prior_n;
    L.all_fast = c->all_keys_fast;
    if (c->cur_prior_n > 0) c->prior_on_device = true;
    c->cur_prior_n = 0;   // (this launch mirrors it)
    c->k1_product_form = k1_product_form(c, L.all_fast);
    if (launch_k1_block(c, L)) return -1;
    if (!defer_ring_event) HIP_TRY(hipEventRecord(c->h_params_ev[c->cur_slot], c->stream));
    c->have_matrices = true;
    c->fold_current = false;  // the folded copy (if any) belongs to the previous matrices
    return 0;
}

// ---- matrices ahead of time (context.hpp, MatrixCache) ----------------------------------------------------------------
// the low-priority stream of the builds ahead of time (creating a stream takes ~12 ms on this runtime: done with the
// context, not inside the first search)
int mc_stream(cafehip_ctx* c)
{
    if (c->mc.stream) return 0;
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = the numerically LARGEST value = the lowest priority
    HIP_TRY(hipStreamCreateWithPriority(&c->mc.stream, hipStreamNonBlocking, lo));
    return 0;
}

// lay the entries out: when the tree, the table's ranges or the error model are set (so that the first search finds the
// store ready), or by the first prefetch after an option changed it.  Returns 1 when the store is off or does not fit.
int mc_prepare(cafehip_ctx* c)
{
    if (c->n_nodes <= 0 || c->M < 0) return 1;
    auto& mc = c->mc;
    if (!mc.e.empty()) return 0;
    if (mc.broken || mc.want_entries <= 0) return 1;
    mc.kpe = std::max(c->n_nodes - 1, 1);
    const size_t per_entry = (size_t)mc.kpe * c->KP * c->LD * sizeof(double) * (c->d_err ? 2 : 1);
    int n = mc.want_entries;
    while (n > 0 && (size_t)n * per_entry > mc.max_bytes) --n;
    if (n < 3) {   // (the bound entry + two candidates: fewer is not worth the bookkeeping)
        mc.broken = true;
        return 1;
    }
    if (mc_stream(c)) return -1;
    mc.e.assign(n, cafehip_ctx::MatrixCache::Entry());
    for (auto& e : mc.e) HIP_TRY(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
    mc.bound = -1;
    if (ensure_matrix_storage(c) || ensure_node_key_store(c)) {
        for (auto& e : mc.e) hipEventDestroy(e.ready);
        mc.e.clear();
        mc.broken = true;
        if (ensure_matrix_storage(c) || ensure_node_key_store(c)) return -1;
        return 1;
    }
    return 0;
}

// the entries are dropped (tree or matrix side changed): laid out again by the next prefetch
int mc_drop(cafehip_ctx* c)
{
    auto& mc = c->mc;
    if (mc.e.empty()) {
        mc.pending_sets = 0;
        mc.broken = false;
        return 0;
    }
    if (sync_streams(c)) return -1;
    mc_invalidate(c);
    for (auto& e : mc.e) hipEventDestroy(e.ready);
    mc.e.clear();
    mc.broken = false;
    return 0;
}

bool mc_same_set(const cafehip_ctx* c, const cafehip_ctx::MatrixCache::Entry& e, const double* nl, const double* nm)
{
    // the reference's key test on every non-root node: exact-double equality of lambda and mu (cafe/cafe_tree.c:380-382; the
    // branch lengths are the tree's).  memcmp would tell -0.0 from 0.0 and NaN from itself differently from ==: a NaN rate
    // simply never hits.
    for (int i = 0; i < c->n_nodes; ++i) {
        if (i == c->root) continue;
        if (!(e.nl[i] == nl[i]) || !(e.nm[i] == nm[i])) return false;
    }
    return true;
}

int mc_find(const cafehip_ctx* c, const double* nl, const double* nm)
{
    for (size_t i = 0; i < c->mc.e.size(); ++i)
        if (c->mc.e[i].valid && mc_same_set(c, c->mc.e[i], nl, nm)) return (int)i;
    return -1;
}

// Build the matrices of up to kMaxSets parameter sets into cache entries: ONE K1 launch on the speculation stream (each
// set's keys deduplicated as an evaluation would, stored at the slots of its entry; the launch's first block writes each
// set's node -> slot map into the entry's row of the device store), then the error fold of each entry.  Sets that are
// already there are only touched; a set whose keys do not all take the product form is left to be built on demand (one
// launch has one arithmetic form, and a set must get the form its own evaluation would use).
struct McStaged {
    bool any = false;     // something to launch
    K1Launch L;           // (its stream is chosen by the launcher)
    int ring_slot = 0;
    bool fold = false;
    std::vector<int> entries, nkeys;
};

// host part: pick the entries, stage the keys and the node -> slot maps into a pinned block
int mc_stage(cafehip_ctx* c, int n_sets, const double* node_lambda, const double* node_mu, McStaged& st)
{
    st.any = false;
    auto& mc = c->mc;
    if (n_sets <= 0) return 0;
    if (c->n_nodes <= 0 || c->M < 0) return 0;
    {
        const int rc = mc_prepare(c);
        if (rc != 0) return rc < 0 ? -1 : 0;
    }
    mc.requested += n_sets;
    const int n = c->n_nodes;
    std::vector<char> keep(mc.e.size(), 0);
    if (mc.bound >= 0) keep[mc.bound] = 1;
    std::vector<int> todo;   // request indices to build
    for (int q = 0; q < n_sets && q < kMaxSets; ++q) {
        const double* nl = node_lambda + (size_t)q * n;
        const double* nm = node_mu + (size_t)q * n;
        bool usable = true;
        for (int i = 0; i < n && usable; ++i)
            if (i != c->root && !(c->bl[i] > 0)) usable = false;   // (an evaluation of this tree fails anyway)
        if (!usable) continue;
        const int at = mc_find(c, nl, nm);
        if (at >= 0) {
            keep[at] = 1;
            mc.e[at].tick = ++mc.tick;
            continue;
        }
        bool dup = false;
        for (int t : todo) {
            cafehip_ctx::MatrixCache::Entry probe;
            probe.nl.assign(node_lambda + (size_t)t * n, node_lambda + (size_t)(t + 1) * n);
            probe.nm.assign(node_mu + (size_t)t * n, node_mu + (size_t)(t + 1) * n);
            if (mc_same_set(c, probe, nl, nm)) dup = true;
        }
        if (!dup) todo.push_back(q);
    }
    if (todo.empty()) return 0;
    // victims: invalid entries first, then the least recently used ones -- never the bound entry or one this request names
    std::vector<int> victims;
    for (size_t t = 0; t < todo.size(); ++t) {
        int best = -1;
        for (size_t i = 0; i < mc.e.size(); ++i) {
            if (keep[i]) continue;
            if (best < 0 || (!mc.e[i].valid && mc.e[best].valid) || (mc.e[i].valid == mc.e[best].valid && mc.e[i].tick < mc.e[best].tick)) best = (int)i;
        }
        if (best < 0) break;
        keep[best] = 1;
        victims.push_back(best);
    }
    todo.resize(victims.size());
    if (todo.empty()) return 0;
    const bool fold = c->d_err && c->opt.errfold;
    if (fold && ensure_fold_storage(c)) return -1;

    const int slot = c->ring_pos;
    c->ring_pos = (c->ring_pos + 1) % kParamRing;
    HIP_TRY(hipEventSynchronize(c->h_params_ev[slot]));
    EvalHeader* h = c->h_params[slot];
    KeyParam* keys = eval_keys(h);
    int32_t* node_key = eval_node_key(h, c->key_cap);
    K1Launch& L = st.L;
    L.stream = nullptr;
    L.ep = h;
    L.first_zero = nullptr;
    L.n_prior = 0;
    L.all_fast = true;
    for (int q = 0; q < kMaxSets; ++q) L.set_row[q] = 0;
    int nk = 0, sets = 0;
    auto& built_entries = st.entries;
    auto& built_nkeys = st.nkeys;
    built_entries.clear();
    built_nkeys.clear();
    for (size_t t = 0; t < todo.size(); ++t) {
        const double* nl = node_lambda + (size_t)todo[t] * n;
        const double* nm = node_mu + (size_t)todo[t] * n;
        auto& e = mc.e[victims[t]];
        const int base = (int)(mc.first_slot + (size_t)victims[t] * mc.kpe);
        const int nk0 = nk;
        // (the victim is touched only once the set is accepted -- ADVICE r05: a set skipped for its arithmetic form used to
        // throw a valid entry away for nothing)
        std::vector<int>& set_keys = c->stage_node_key;
        set_keys.assign(n, -1);
        bool all_fast = true;
        auto& kl = c->stage_l;   // this set's distinct (lambda, mu) by key, the branch length in the key itself
        auto& km = c->stage_m;
        kl.clear();
        km.clear();
        for (int i = 0; i < n; ++i) {
            node_key[(size_t)sets * n + i] = 0;
            if (i == c->root) continue;
            const int bl = c->bl_int[i];
            int k = nk0;
            for (; k < nk; ++k)
                if (keys[k].bl == bl && kl[k - nk0] == nl[i] && km[k - nk0] == nm[i]) break;
            if (k == nk) {
                fill_key(c, keys[k], bl, nl[i], nm[i], base + (k - nk0));
                kl.push_back(nl[i]);
                km.push_back(nm[i]);
                if (keys[k].mode >= 2 && !keys[k].fast_ok) all_fast = false;
                ++nk;
            }
            set_keys[i] = base + (k - nk0);
            node_key[(size_t)sets * n + i] = base + (k - nk0);
        }
        if (k1_product_form(c, all_fast) != k1_product_form(c, true)) {
            nk = nk0;   // this set's own evaluation would run another arithmetic form than the launch: built on demand
            continue;
        }
        if (e.valid) ++mc.evicted;
        e.valid = false;
        e.node_key = set_keys;
        e.nl.assign(nl, nl + n);
        e.nm.assign(nm, nm + n);
        e.nkeys = nk - nk0;
        e.folded = false;
        e.ready_known = false;
        L.set_row[sets] = kMaxSets + victims[t];
        built_entries.push_back(victims[t]);
        built_nkeys.push_back(nk - nk0);
        ++sets;
    }
    h->nkeys = nk;
    h->n_sets = sets;
    h->n_nodes = n;
    h->key_cap = c->key_cap;
    if (sets == 0) return 0;
    L.nkeys = nk;
    L.n_sets = sets;
    st.ring_slot = slot;
    st.fold = fold;
    st.any = true;
    return 0;
}

// behind the launch that builds the staged sets on `stream`: the ring slot's event, the error folds, the entries' state.
// same_stream_as_readers: every launch that will read the entries is queued on `stream` too (no event needed).
int mc_finish(cafehip_ctx* c, McStaged& st, hipStream_t stream, bool same_stream_as_readers)
{
    auto& mc = c->mc;
    HIP_TRY(hipEventRecord(c->h_params_ev[st.ring_slot], stream));
    ++mc.launches;
    for (size_t t = 0; t < st.entries.size(); ++t) {
        auto& e = mc.e[st.entries[t]];
        if (st.fold) {
            if (launch_fold_slots(c, stream, mc.first_slot + (size_t)st.entries[t] * mc.kpe, st.nkeys[t])) return -1;
            e.folded = true;
        }
        if (same_stream_as_readers && stream == c->stream) e.ready_known = true;
        else HIP_TRY(hipEventRecord(e.ready, stream));
        e.valid = true;
        e.tick = ++mc.tick;
        ++mc.built;
    }
    st.any = false;
    return 0;
}

// stage + one K1 launch + finish.  Where the build runs (option prefetch_where): 0 = on the second, low-priority stream at
// once -- beside the pruning of the evaluation just launched; 1 = on the context's own stream, i.e. behind whatever is
// queued there; 2 = on the second stream but not before the context's stream has drained to this point
int mc_build(cafehip_ctx* c, int n_sets, const double* node_lambda, const double* node_mu)
{
    auto& mc = c->mc;
    McStaged st;
    if (mc_stage(c, n_sets, node_lambda, node_mu, st)) return -1;
    if (!st.any) return 0;
    const int where = c->opt.prefetch_where;
    hipStream_t build_stream = (where == 1 || where == 3) ? c->stream : mc.stream;
    if (where == 2) {
        if (!mc.chain_end) HIP_TRY(hipEventCreateWithFlags(&mc.chain_end, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(mc.chain_end, c->stream));
        HIP_TRY(hipStreamWaitEvent(mc.stream, mc.chain_end, 0));
    }
    st.L.stream = build_stream;
    st.L.kpb = build_stream == mc.stream ? c->opt.prefetch_kpb : 0;
    if (launch_k1_block(c, st.L)) return -1;
    return mc_finish(c, st, build_stream, build_stream == c->stream);
}

// a request parked by cafehip_prefetch_matrices(..., CAFEHIP_PREFETCH_BEHIND_NEXT_EVALUATION): issued once the evaluation's
// own launches are in the queue, so that they are not delayed by the host work of staging the candidates
int mc_issue_pending(cafehip_ctx* c)
{
    auto& mc = c->mc;
    if (mc.pending_sets <= 0) return 0;
    const int n = mc.pending_sets;
    mc.pending_sets = 0;
    return mc_build(c, n, mc.pending_l.data(), mc.pending_m.data());
}

// An evaluation of (node_lambda, node_mu) whose matrices are in the cache: bind the nodes to them instead of building.
// Returns the entry or -1.
int mc_bind(cafehip_ctx* c, const double* node_lambda, const double* node_mu, const double* prior)
{
    auto& mc = c->mc;
    if (mc.e.empty()) return -1;
    // what K1 would have done besides the matrices must already hold: the prior on the device is this one, the
    // first-zero word is reset (the score kernels leave it so)
    if (!c->fz_clean || !c->prior_on_device || !prior || (int)c->prior_seen.size() != c->R ||
        memcmp(c->prior_seen.data(), prior, sizeof(double) * c->R) != 0) {
        return -1;
    }
    const int at = mc_find(c, node_lambda, node_mu);
    if (at < 0) {
        ++mc.misses;
        return -1;
    }
    auto& e = mc.e[at];
    if (c->d_err && c->opt.errfold && !e.folded) return -1;
    if (!e.ready_known) {
        const hipError_t q = hipEventQuery(e.ready);
        if (q == hipSuccess) {
            e.ready_known = true;
        } else if (q == hipErrorNotReady) {
            if (hipStreamWaitEvent(c->stream, e.ready, 0) != hipSuccess) return -1;   // the build is still running: the pruning waits for it
            ++mc.waited;
        } else {
            return -1;
        }
    }
    ++mc.hits;
    e.tick = ++mc.tick;
    mc.bound = at;
    c->cur_node_key = c->d_node_key + (size_t)(kMaxSets + at) * c->n_nodes;
    c->node_key = e.node_key;
    c->nkeys = e.nkeys;
    c->cur_sets = 1;
    c->have_matrices = true;
    c->fold_current = e.folded;
    return at;
}

