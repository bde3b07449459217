// cafehip.hip -- context (context.hpp) and C ABI of the MI355X (gfx950) engine for CAFE's per-family likelihood hot path.
// See include/cafehip.h for the boundary and DESIGN.md for the layout.  The kernels live in their own translation
// units (kernels.hpp):
//
//   K1  k1_matrices.hip     birth-death transition matrices for every unique (int branch length, lambda, mu) key
//                           of one evaluation == compute_birthdeath_rates, libtree/birthdeath.c:238-286;
//                           k1e_fold_error: error model folded into those matrices, cafe/cafe_tree.c:196-203
//   K2  k2_walk16.hip,      post-order pruning of ALL families in one launch + the per-family posterior
//       k2_walk4.hip,       == compute_tree_likelihoods + compute_posterior, cafe/cafe_tree.c:191-323,
//       k2c_tables.hip,     cafe/lambda.cpp:657-689 (k2_mfma.hpp); factor tables of compressed subtrees;
//       k_misc.hip          row-per-thread fallback k2_prune_v1
//   K3  k_misc.hip          per-chunk sums of log max-posterior in family order and the first zero-likelihood
//                           family == get_posterior, cafe/lambda.cpp:691-724
//   K4  k_misc.hip          max-product walk + backtrack == cafe_tree_viterbi, cafe/viterbi.cpp:208-351
//
// gfx950 only.  No CPU fallback: every entry point fails if the device work fails.
#include "context.hpp"
#include "exp_like_host.hpp"

namespace {

void free_family_buffers(cafehip_ctx* c)
{
    hipFree(c->d_counts);
    hipFree(c->d_fam2u);
    hipFree(c->d_max_lik);
    hipFree(c->d_max_post);
    hipFree(c->d_argmax);
    hipFree(c->d_chunk_sums);
    c->d_counts = c->d_fam2u = c->d_argmax = nullptr;
    c->d_max_lik = c->d_max_post = c->d_chunk_sums = nullptr;
}

// dynamic LDS above the default limit must be granted per kernel and per device
int grant_lds(cafehip_ctx* c, const void* fn, size_t lds, size_t default_limit)
{
    if (lds <= default_limit) return 0;
    size_t& have = c->lds_attr[fn];
    if (lds <= have) return 0;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    have = lds;
    return 0;
}

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
    hipFree(c->d_PT);
    c->d_PT = nullptr;
    const size_t keep = std::max(need_keys, c->pt_keys_cap);
    const size_t bytes = (keep + mc_slots(c)) * (size_t)c->KP * c->LD * sizeof(double);
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
        P.grid = dim3((c->S + 16 * K1Q - 1) / (16 * K1Q), (c->S + 15) / 16, (L.nkeys + kpb - 1) / kpb);
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
        if (e.valid) ++mc.evicted;
        e.valid = false;
        const int base = (int)(mc.first_slot + (size_t)victims[t] * mc.kpe);
        const int nk0 = nk;
        e.node_key.assign(n, -1);
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
            e.node_key[i] = base + (k - nk0);
            node_key[(size_t)sets * n + i] = base + (k - nk0);
        }
        if (k1_product_form(c, all_fast) != k1_product_form(c, true)) {
            nk = nk0;   // this set's own evaluation would run another arithmetic form than the launch: built on demand
            continue;
        }
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

constexpr size_t kV1LdsLarge = 150 * 1024;   // node-vector slots of the row-per-thread kernel (one workgroup per CU)

int launch_k2_v1(cafehip_ctx* c, K2Args& a, int n_items)
{
    if (n_items <= 0) return 0;
    const int slots = c->sched.n_slots + (c->d_err ? 1 : 0);
    const int rows_max = std::max(c->C, c->R);
    int block = ((rows_max + 63) / 64) * 64;
    if (block > 1024)
        return fail("matrix side %d exceeds the 1024 rows this kernel handles", rows_max);
    // (the reference-arithmetic form keeps a separate product and sum per term: 16 families per workgroup spilled 15
    // registers to scratch -- the only spilling kernel of the library, VERDICT r04 -- so it runs 8)
    int nf = c->opt.k2 == 2 ? 8 : 16;
    size_t lds = 0;
    for (; nf >= 1; nf >>= 1) {
        lds = (size_t)slots * nf * c->LDv * sizeof(double) + (size_t)nf * (c->n_leaves + 1) * sizeof(int);
        if (lds <= kV1LdsLarge) break;
    }
    if (nf < 1)
        return fail("tree needs %d live node vectors of %d doubles: does not fit %zu B of LDS",
                    slots, c->LDv, kV1LdsLarge);
    c->k2_nf = nf;
    c->k2_block = block;
    c->k2_lds = lds;
    a.n_slots = c->sched.n_slots;
    const void* fn = k2_v1_kernel(nf, c->opt.k2 == 2);
    if (grant_lds(c, fn, lds, 64 * 1024)) return -1;
    return launch_kernel(fn, dim3((n_items + nf - 1) / nf), dim3(block), lds, c->stream, a);
}


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

// wave rows / row tiles per wave of k2c_nodes for this matrix side: one wave per row tile up to 16 waves
// (0: matrix too large, no compression)
int k2c_wave_rows(const cafehip_ctx* c, int* nrt_w)
{
    const int RT = (c->C + 15) / 16;
    const int wr = std::min(RT, 16);
    *nrt_w = (RT + wr - 1) / wr;
    return *nrt_w <= 2 ? wr : 0;   // (matrix sides up to 512; beyond, the plain walk)
}

// (Re)build the plan from the tree and the unique rows.  A node is compressed when both children are leaves or
// compressed and its distinct states number at most `compress_theta` of the unique rows (default by table size
// and matrix side, see below); option compress=0 disables.  Tables with fewer than 64 unique rows (`compress_min`)
// are left alone.
int rebuild_compression(cafehip_ctx* c)
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
    }
    if (c->opt.compress_theta >= 0) theta = std::min(c->opt.compress_theta, 1.0);
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
    if (elems >= ((size_t)1 << 31) || n_idx >= ((size_t)1 << 31)) { p = cafehip_ctx::CompressPlan(); return 0; }
    p.table_elems = elems;
    // tiles, level by level (children's tables are complete before a level starts)
    p.level_first.assign(1, 0);
    p.level_nft.clear();
    for (int l = 1; l <= n_levels; ++l) {
            // 16 states per tile: 32- and 64-state tiles (a half / a quarter of the workgroups and of the matrix re-reads)
        // measured 2-30 % slower at every bench shape -- fewer, longer workgroups fill the chip worse
        const int nft = 1;
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
    int nrt_w = 0;
    const int wr = k2c_wave_rows(c, &nrt_w);
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
        if (launch_k2c_inst(c, k2c_kernel(nft, nrt_w, c->opt.k2c_batch != 0), nft, a, n_tiles, n_sets, 64 * wr)) return -1;
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
    const void* fn = k2_mfma16_kernel(nft_w, nrt_w);
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
    const void* fn = k2_mfma4_kernel(G, nrt_w);
    if (!fn) return fail("unsupported 4x4 wave grid G=%d NRT_W=%d", G, nrt_w);
    if (grant_lds(c, fn, lds, 64 * 1024)) return -1;
    if (k2_fit_grid(c, fn, a, &grid, block, lds)) return -1;
    return launch_kernel(fn, dim3(grid, a.n_sets), dim3(block), lds, c->stream, a);
}

// measured wave-grid choices of this process, by problem shape
constexpr int kTuneReps = 4;
constexpr int kTuneRounds = 5;   // round 0 warm-up, round 1 every grid, rounds 2-4 those within 5 % of the best (minimum kept):
                                 // with one re-timing the choice between two grids 3 % apart flipped in one run out of five
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
                } while (t.round >= 2 && t.round < kTuneRounds && t.best_ms[t.cur] > 1.05f * best);
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
        reps = warm < 0.25f ? kTuneReps : (warm < 1.0f ? 2 : 1);
    }
    if (tuning_launch) c->tune.reps_launched = reps;
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

int check_ready(cafehip_ctx* c)
{
    if (!c) return fail("null context");
    if (c->n_nodes <= 0) return fail("cafehip_set_tree has not been called");
    if (c->M < 0) return fail("cafehip_set_families has not been called");
    if (c->n_leaves != (c->n_nodes + 1) / 2)
        return fail("count table has %d columns but the tree has %d leaves", c->n_leaves,
                    (c->n_nodes + 1) / 2);
    return 0;
}

int ensure_output_sets(cafehip_ctx* c, int n_sets)
{
    if (n_sets <= c->out_sets) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(c->d_max_lik);
    hipFree(c->d_max_post);
    hipFree(c->d_argmax);
    hipFree(c->d_chunk_sums);
    c->d_max_lik = c->d_max_post = c->d_chunk_sums = nullptr;
    c->d_argmax = nullptr;
    c->out_sets = 0;
    const size_t fu = (size_t)std::max(c->Fu, 1) * n_sets;
    HIP_TRY(hipMalloc(&c->d_max_lik, fu * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_max_post, fu * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_argmax, fu * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->d_chunk_sums, (size_t)std::max(c->n_chunks, 1) * n_sets * sizeof(double)));
    const size_t need = (size_t)std::max(c->n_chunks, 1) * n_sets;
    if (!c->h_result || need > c->h_result_chunks) {
        hipHostFree(c->h_result);
        c->h_result = nullptr;
        const size_t bytes = sizeof(HostResult) + need * sizeof(double);
        HIP_TRY(hipHostMalloc((void**)&c->h_result, bytes, hipHostMallocMapped | hipHostMallocCoherent));
        memset((void*)c->h_result, 0, bytes);
        c->h_result_chunks = need;
    }
    c->out_sets = n_sets;
    return 0;
}

}  // namespace

// (declared in context.hpp: the multi-GPU entry points of cafehip_comm.hip run the same evaluation)
namespace {

// whether the pruning launch of an objective evaluation has stopped trying wave grids (a pre-armed chain repeats the launch
// as it is)
bool k2_settled(const cafehip_ctx* c)
{
    if (c->opt.k2 != 0) return true;
    if (!c->opt.k2tune || c->opt.mfma != 0 || c->opt.have_cfg16 || c->opt.have_cfg4) return true;
    return c->tune.n_items == c->Fu && c->tune.locked >= 0 && !c->tune.pending;
}

// K1 (unless the nodes are bound to matrices built ahead) -> error fold -> table levels -> walk -> score kernel of the
// evaluation staged in c->cur_params, on the context's stream.  *seq_out: the sequence number its score kernel publishes.
int launch_chain(cafehip_ctx* c, double* d_chunk_sums, int32_t* d_first_zero, bool host_out, int n_sets, bool direct_exchange, bool bound,
                 int32_t* seq_out)
{
    RingGuard ring(c);
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if (!bound) {
        ring.arm();   // from here on the slot's event is recorded on every way out
        if (launch_k1(c, d_first_zero, true)) return -1;
        if (launch_error_fold(c)) return -1;
    }
    c->fz_clean = false;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    K2Args a;
    fill_common_k2(c, a);
    a.counts = c->d_counts;
    a.Fu = c->Fu;
    a.max_lik = c->d_max_lik;
    a.argmax = c->d_argmax;
    a.max_post = c->d_max_post;
    if (c->d_err) {
        a.err = c->d_err;
        a.err_ld = c->err_mfs + 1;
        a.leaf_has_err = c->d_leaf_has_err;
    }
    {
        // the objective path walks the reduced tree when the table compresses (matrix-core kernels; an error model
        // only in its folded form, so that every leaf stays a column gather)
        const bool use_c = c->cp.valid && c->opt.k2 == 0 && (!c->d_err || c->fold_current);
        c->issued_tables = 0;
        if (use_c && launch_compressed_levels(c, n_sets)) return -1;
        c->ev_mid_used = false;
        if (use_c && c->timing) {
            if (!c->ev_mid) HIP_TRY(hipEventCreate(&c->ev_mid));
            HIP_TRY(hipEventRecord(c->ev_mid, c->stream));
            c->ev_mid_used = true;
        }
        c->walk_compressed = use_c;
        const int rc = launch_k2(c, a, c->Fu, n_sets);
        c->walk_compressed = false;
        if (rc) return -1;
        c->last_compressed = use_c && c->k2_used_mfma;
    }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    if (direct_exchange) {
        // sharded evaluation, direct exchange: the score kernel stores this rank's packed row into every rank's
        // exchange buffer and hands all rows to the host (k3_score_x)
        CommLink& L = *c->link;
        K3xArgs x;
        memset(&x, 0, sizeof x);
        x.max_post_u = c->d_max_post;
        x.max_lik_u = c->d_max_lik;
        x.fam2u = c->F == c->Fu ? nullptr : c->d_fam2u;   // (no duplicate rows: family i is unique row i)
        x.F = c->F;
        x.Fu = c->Fu;
        x.first_zero = d_first_zero;
        x.host = c->h_result;
        x.arrive = c->d_arrive;
        x.seq = ++c->host_seq;
        if (seq_out) *seq_out = x.seq;
        x.rank = L.rank;
        x.world = L.world;
        x.slots = c->x_slots;
        x.xseq = c->x_seq + 1;
        const int parity = (int)(x.xseq & 1);
        for (int r = 0; r < L.world; ++r) {
            x.rows[r] = CommLink::rows_of(L.peer_xbuf[r], parity);
            x.flags[r] = reinterpret_cast<unsigned long long*>(CommLink::flags_of(L.peer_xbuf[r], parity));
        }
        // a rank that died must not hang the others' GPUs, and no GPU sits in one kernel for long: the in-kernel wait is
        // one slice (<= 1 s, in ticks of the 100 MHz wall clock); a rank that is merely late (it wrote a report, loaded
        // a table) is waited for by the HOST, slice after slice, for comm_timeout_s (cafehip_eval_posterior_sharded).
        // Each rank decides alone when to give up -- nothing here needs the ranks to agree.
        x.timeout_ticks = (long long)(x_wait_slice_s() * 1e8);
        if (launch_kernel(k3x_kernel(), dim3(std::max(c->n_chunks, 1)), dim3(CAFEHIP_CHUNK), 0, c->stream, x)) return -1;
        c->x_seq = x.xseq;   // the launch went out: only now is the number taken
        c->x_last = x;
        c->fz_clean = d_first_zero == c->d_first_zero;   // (its last block reads the word with an exchange)
    } else if (c->n_chunks > 0) {
        K3Args k3{c->d_max_post, c->d_max_lik, c->F == c->Fu ? nullptr : c->d_fam2u, c->F, c->Fu, d_chunk_sums, d_first_zero, nullptr, nullptr, 0};
        if (host_out) {
            k3.host = c->h_result;
            k3.arrive = c->d_arrive;
            k3.seq = ++c->host_seq;
            if (seq_out) *seq_out = k3.seq;
        }
        // Candidates announced for the NEXT evaluation (cafehip_prefetch_matrices, parked): their matrices are built by the
        // trailing blocks of the score kernel's own launch (k3_score_then_k1_rb), i.e. while the score travels to the host
        // and the optimiser decides -- the chip is idle then, and the next chain queues behind this launch anyway.  Staging
        // them first costs the host a few microseconds the device spends in the walk.
        McStaged st;
        if (c->opt.prefetch_where == 3 && c->mc.pending_sets > 0 && host_out && n_sets == 1) {
            const int n = c->mc.pending_sets;
            c->mc.pending_sets = 0;
            if (mc_stage(c, n, c->mc.pending_l.data(), c->mc.pending_m.data(), st)) return -1;
        }
        bool fused = false;
        if (st.any) {
            K1Plan P;
            st.L.stream = c->stream;
            if (plan_k1_block(c, st.L, P)) return -1;
            if (P.register_blocked) {
                K3K1Args f;
                f.k3 = k3;
                f.k1 = P.a;
                f.k3_blocks = c->n_chunks;
                f.gx = (int)P.grid.x;
                f.gy = (int)P.grid.y;
                const void* fn = k3_then_k1_rb_kernel();
                if (grant_lds(c, fn, P.lds, 48 * 1024)) return -1;
                if (launch_kernel(fn, dim3(c->n_chunks + P.grid.x * P.grid.y * P.grid.z), dim3(CAFEHIP_CHUNK), P.lds, c->stream, f)) return -1;
                fused = true;
            }
        }
        if (!fused && launch_kernel(k3_kernel(host_out), dim3(c->n_chunks, n_sets), dim3(CAFEHIP_CHUNK), 0, c->stream, k3)) return -1;
        c->fz_clean = host_out && d_first_zero == c->d_first_zero;
        if (st.any) {
            if (!fused && launch_k1_block(c, st.L)) return -1;   // (another arithmetic form: its own launch behind the score kernel)
            if (mc_finish(c, st, c->stream, true)) return -1;
        }
    }
    if (!bound && ring.record_now()) return -1;   // (deferred from launch_k1)
    // candidates announced for the NEXT evaluation: their matrices are built now, beside this evaluation's pruning
    if (mc_issue_pending(c)) return -1;
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        c->timing_pending = true;
    }
    return 0;
}

// Queue the NEXT evaluation's launches behind a gate (context.hpp, Armed).  The block its K1 will read is reserved now and
// filled with a copy of the current evaluation's parameters, so that the chain is a valid evaluation whatever happens.
int arm_next(cafehip_ctx* c, double* d_chunk_sums, int32_t* d_first_zero)
{
    if (!c->opt.prearm || c->timing || c->stream != c->own_stream || c->n_chunks <= 0 || c->nkeys <= 0 || !k2_settled(c)) return 0;
    // worth a microsecond per evaluation (115.3 -> 114.3 us at configs[1], profiles/r05/prearm_ab.txt): only where an evaluation
    // is short -- a chain let go unused repeats a whole evaluation, which a table that fills the chip should not pay
    if (c->k2_grid <= 0 || c->k2_grid > 2 * std::max(c->n_cu, 1)) return 0;
    if (c->mc.pending_sets > 0 || c->mc.requested != c->mc_requested_seen) return 0;   // somebody announces sets: the store serves them
    if (!c->h_gate) {
        HIP_TRY(hipHostMalloc((void**)&c->h_gate, 32 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->h_gate, 0, 32 * sizeof(unsigned long long));
    }
    const int slot = c->ring_pos;
    c->ring_pos = (c->ring_pos + 1) % kParamRing;
    HIP_TRY(hipEventSynchronize(c->h_params_ev[slot]));
    memcpy(c->h_params[slot], c->cur_params, eval_prior_offset(c->key_cap, c->n_nodes));   // header, keys, node -> matrix maps
    GateArgs g;
    g.flag = c->h_gate;
    g.outcome = c->h_gate + 16;
    g.want = ++c->gate_seq;
    g.timeout_ticks = 2000000;   // 20 ms of the 100 MHz clock
    if (launch_kernel(gate_kernel(), dim3(1), dim3(64), 0, c->stream, g)) return -1;
    c->cur_params = c->h_params[slot];
    c->cur_slot = slot;
    c->cur_prior_n = 0;
    int32_t seq = 0;
    if (launch_chain(c, d_chunk_sums, d_first_zero, true, 1, false, false, &seq)) {
        // (the gate is in the queue: let it go, whatever made it behind it runs on the copied block)
        __atomic_store_n(&c->h_gate[0], g.want, __ATOMIC_RELEASE);
        return -1;
    }
    c->armed.on = true;
    c->armed.slot = slot;
    c->armed.nkeys = c->nkeys;
    c->armed.all_fast = c->all_keys_fast;
    c->armed.seq = seq;
    c->armed.gate = g.want;
    return 0;
}

}  // namespace

int cafehip_impl::eval_device(cafehip_ctx* c, const double* node_lambda, const double* node_mu, const double* prior, double* d_chunk_sums,
                              int32_t* d_first_zero, bool host_out, int n_sets, bool direct_exchange)
{
    if (check_ready(c)) return -1;
    HIP_TRY(hipSetDevice(c->device));
    if (n_sets > 1 && ensure_output_sets(c, n_sets)) return -1;
    if (n_sets > 1 && d_chunk_sums == nullptr) {
        d_chunk_sums = c->d_chunk_sums;   // (re)allocated above
    }
    // the matrices of this set may already be on the device (cafehip_prefetch_matrices): then the nodes are bound to them
    // and the chain starts at the pruning -- no staging, no K1, no fold.  Only for the calls whose score kernel leaves the
    // first-zero word reset (the synchronous and the direct-exchange paths), one set at a time.
    bool staged = false;
    if (c->armed.on) {
        // A chain for exactly this call is waiting behind its gate: stage the parameters into the block it reads and let it
        // go with one store.  It fits if the evaluation has the shape it was armed with (as many distinct matrices in the same
        // arithmetic form, the prior already on the device) and the gate has not given up meanwhile.
        const bool gate_alive = __atomic_load_n(&c->h_gate[16], __ATOMIC_ACQUIRE) != ((c->armed.gate << 2) | 2ull);
        if (n_sets == 1 && host_out && !direct_exchange && d_first_zero == c->d_first_zero && !c->timing && gate_alive &&
            c->mc.pending_sets == 0 && c->mc.requested == c->mc_requested_seen) {
            if (stage_params(c, node_lambda, node_mu, prior, 1, c->armed.slot)) {
                disarm(c);
                return -1;
            }
            staged = true;
            if (c->nkeys == c->armed.nkeys && c->all_keys_fast == c->armed.all_fast && c->cur_prior_n == 0) {
                std::atomic_thread_fence(std::memory_order_release);
                __atomic_store_n(&c->h_gate[0], c->armed.gate, __ATOMIC_RELEASE);
                c->armed.on = false;
                ++c->prearm_used;
                c->eval_seq = c->armed.seq;
                c->released_gate = c->armed.gate;
                c->have_matrices = true;
                c->mc.bound = -1;
                if (arm_next(c, d_chunk_sums, d_first_zero)) return -1;
                return 0;
            }
            // another shape: the armed chain runs on what the block holds now (every slot and count in it stays in range) and
            // its result is ignored; this evaluation is launched the ordinary way behind it, from the same block
        }
        disarm(c);
    }
    const bool may_bind = !staged && n_sets == 1 && (host_out || direct_exchange) && d_first_zero == c->d_first_zero && !c->mc.e.empty();
    const bool bound = may_bind && mc_bind(c, node_lambda, node_mu, prior) >= 0;
    if (!bound && !staged && stage_params(c, node_lambda, node_mu, prior, n_sets)) return -1;
    int32_t seq = c->host_seq;
    if (launch_chain(c, d_chunk_sums, d_first_zero, host_out, n_sets, direct_exchange, bound, &seq)) return -1;
    c->eval_seq = seq;
    c->released_gate = 0;
    if (!bound && n_sets == 1 && host_out && !direct_exchange && d_first_zero == c->d_first_zero && arm_next(c, d_chunk_sums, d_first_zero)) return -1;
    c->mc_requested_seen = c->mc.requested;
    return 0;
}

void cafehip_impl::disarm(cafehip_ctx* c)
{
    if (!c || !c->armed.on) return;
    // let it go: its K1 reads the block it was armed with (a copy of the previous evaluation's parameters), the chain
    // repeats that evaluation and publishes a sequence number nobody waits for
    std::atomic_thread_fence(std::memory_order_release);
    __atomic_store_n(&c->h_gate[0], c->armed.gate, __ATOMIC_RELEASE);
    c->armed.on = false;
    ++c->prearm_wasted;
}

bool cafehip_impl::armed_chain_expired(cafehip_ctx* c)
{
    if (!c->released_gate) return false;
    const unsigned long long got = __atomic_load_n(&c->h_gate[16], __ATOMIC_ACQUIRE);
    const bool expired = got != ((c->released_gate << 2) | 1ull);
    c->released_gate = 0;
    if (expired) ++c->prearm_expired;
    return expired;
}

// elapsed times of the last evaluation's three launches (blocks until its last event has completed)
int cafehip_impl::collect_kernel_ms(cafehip_ctx* c)
{
    if (!c->timing || !c->timing_pending) return 0;
    HIP_TRY(hipEventSynchronize(c->ev[3]));
    for (int i = 0; i < 3; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->last_ms[i] = ms;
    }
    c->last_tables_ms = 0;
    if (c->ev_mid_used) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev_mid));
        c->last_tables_ms = ms;
    }
    c->timing_pending = false;
    return 0;
}

namespace {

// ---- run-time switches ---------------------------------------------------------------------------------------
// (name, what it selects) -- cafehip_set_option; the same names upper-cased behind CAFEHIP_ are read from the
// environment ONCE, when the context is created (tools/ sweeps), never during an evaluation
const char* const kOptionNames[] = {"compress", "compress_theta", "compress_min", "compress_max_level", "compress_drop_top", "errfold", "errband", "k1", "k1kpb", "k2", "mfma",
                                    "k2cfg", "k2cfg4", "k2tune", "k2tune_log", "k2slots", "ldspark", "vitlds", "k2c_batch",
                                    "batch_trim", "batch_lockstep", "walk_lockstep", "batch_lockstep_slack", "exp_like_host", "matrix_cache",
                                    "matrix_cache_mb", "prefetch_where", "prefetch_kpb", "prearm", "comm"};

int set_option(cafehip_ctx* c, const std::string& key, const std::string& val)
{
    auto& o = c->opt;
    const int iv = atoi(val.c_str());
    bool replan = false;
    if (key == "compress") { o.compress = iv != 0; replan = true; }
    else if (key == "compress_theta") { o.compress_theta = val.empty() ? -1.0 : std::min(std::max(atof(val.c_str()), 0.0), 1.0); replan = true; }
    else if (key == "compress_min") { o.compress_min = iv; replan = true; }
    else if (key == "compress_max_level") { o.compress_max_level = iv; replan = true; }
    else if (key == "compress_drop_top") { o.compress_drop_top = iv != 0; replan = true; }
    else if (key == "errfold") o.errfold = iv != 0;
    else if (key == "errband") {
        o.errband = iv != 0;
        c->err_banded = c->err_mfs >= 0 && c->err_band_width <= 16 && o.errband;
    } else if (key == "k1") {
        if (val == "exact") o.k1 = 1;
        else if (val == "perterm") o.k1 = 2;
        else if (val.empty() || val == "auto" || val == "rb") o.k1 = 0;
        else return fail("option k1: auto | exact | perterm, got '%s'", val.c_str());
    } else if (key == "k1kpb") o.k1_kpb = std::max(1, iv);
    else if (key == "k2") {
        if (val == "v1") o.k2 = 1;
        else if (val == "v1ref") o.k2 = 2;   // ... in the reference's arithmetic (separate multiply and add per term)
        else if (val.empty() || val == "auto" || val == "mfma") o.k2 = 0;
        else return fail("option k2: auto | mfma | v1 | v1ref, got '%s'", val.c_str());
    } else if (key == "mfma") {
        if (val == "4") o.mfma = 4;
        else if (val == "16") o.mfma = 16;
        else if (val.empty() || val == "auto") o.mfma = 0;
        else return fail("option mfma: auto | 4 | 16, got '%s'", val.c_str());
    } else if (key == "k2cfg" || key == "k2cfg4") {
        int* dst = key == "k2cfg" ? o.cfg16 : o.cfg4;
        bool& have = key == "k2cfg" ? o.have_cfg16 : o.have_cfg4;
        have = false;
        if (!val.empty()) {
            if (sscanf(val.c_str(), "%d,%d,%d,%d", dst, dst + 1, dst + 2, dst + 3) != 4)
                return fail("option %s: \"a,b,wf,wr\", got '%s'", key.c_str(), val.c_str());
            have = true;
        }
    } else if (key == "k2tune") o.k2tune = iv != 0;
    else if (key == "k2tune_log") o.k2tune_log = iv != 0;
    else if (key == "k2slots") o.k2slots = iv != 0;
    else if (key == "ldspark") o.ldspark = val.empty() ? -1 : iv;
    else if (key == "vitlds") o.vitlds = iv != 0;
    else if (key == "k2c_batch") o.k2c_batch = iv != 0;
    else if (key == "batch_trim") o.batch_trim = iv != 0;
    else if (key == "batch_lockstep") o.batch_lockstep = iv != 0;
    else if (key == "walk_lockstep") o.walk_lockstep = iv != 0;
    else if (key == "batch_lockstep_slack") o.batch_lockstep_slack = std::min(std::max(iv, 0), 100);
    else if (key == "exp_like_host") o.exp_like_host = iv != 0;
    else if (key == "prearm") { disarm(c); o.prearm = iv != 0; return 0; }
    else if (key == "prefetch_kpb") o.prefetch_kpb = std::max(iv, 0);
    else if (key == "prefetch_where") o.prefetch_where = std::min(std::max(iv, 0), 3);
    else if (key == "matrix_cache" || key == "matrix_cache_mb") {
        // entries of the matrices-ahead-of-time store (0: cafehip_prefetch_matrices is ignored) / its size limit
        HIP_TRY(hipSetDevice(c->device));
        if (mc_drop(c)) return -1;
        if (key == "matrix_cache") c->mc.want_entries = val.empty() ? 12 : std::min(std::max(iv, 0), 64);
        else c->mc.max_bytes = (size_t)std::max(iv, 0) << 20;
        if (c->M >= 0 && c->n_nodes > 0 && (ensure_matrix_storage(c) || ensure_node_key_store(c))) return -1;
        return 0;
    }
    else if (key == "comm") {
        if (val == "rccl") c->comm_mode = 1;
        else if (val == "direct") c->comm_mode = 2;
        else if (val.empty() || val == "auto") c->comm_mode = 0;
        else return fail("option comm: auto | direct | rccl, got '%s'", val.c_str());
        return 0;
    }
    else return fail("unknown option '%s'", key.c_str());
    c->tune.n_items = -1;   // the wave grid is measured again under the new switches
    mc_invalidate(c);       // ... and matrices built ahead of time may have been built under the old ones
    if (replan && c->M >= 0 && c->n_nodes > 0) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (rebuild_compression(c)) return -1;
    }
    return 0;
}

void options_from_environment(cafehip_ctx* c)
{
    for (const char* name : kOptionNames) {
        std::string env = "CAFEHIP_";
        for (const char* p = name; *p; ++p) env += (char)toupper((unsigned char)*p);
        if (const char* v = getenv(env.c_str()))
            if (set_option(c, name, v) != 0) fprintf(stderr, "cafehip: %s=%s ignored: %s\n", env.c_str(), v, g_err.c_str());
    }
}

}  // namespace

static int launch_k4_nf(cafehip_ctx* c, int nf, const K4Args& a, int block, size_t lds)
{
    const void* fn = k4_kernel(nf);
    if (grant_lds(c, fn, lds, 48 * 1024)) return -1;
    return launch_kernel(fn, dim3((a.B + nf - 1) / nf), dim3(block), lds, c->stream, a);
}


// ====================================================================================
// C ABI
// ====================================================================================
extern "C" {

int cafehip_abi_version(void) { return 1; }

const char* cafehip_last_error(void) { return g_err.c_str(); }

int cafehip_create(cafehip_ctx** out, int device_id)
{
    if (!out) return fail("null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail("no HIP device available (%s): cafehip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail("device %d out of range [0,%d)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    (void)host_exp_variant_once();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    cafehip_ctx* c = new cafehip_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    c->lds_limit = (int)prop.sharedMemPerBlock;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && v > c->lds_limit)
            c->lds_limit = v;
        // gfx950 has 160 KiB per CU; a single workgroup may use all of it
        if (strstr(prop.gcnArchName, "gfx950") && c->lds_limit < 160 * 1024) c->lds_limit = 160 * 1024;
    }
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipMalloc(&c->d_prior, kMaxPrior * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_logprior, kMaxPrior * sizeof(double)));
    for (int i = 0; i < kParamRing; ++i) HIP_TRY(hipEventCreateWithFlags(&c->h_params_ev[i], hipEventDisableTiming));
    options_from_environment(c);
    if (c->mc.want_entries > 0 && mc_stream(c)) return -1;
    for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreate(&c->ev[i]));
    HIP_TRY(hipMalloc(&c->d_first_zero, kMaxSets * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->d_arrive, sizeof(int32_t)));
    HIP_TRY(hipMemset(c->d_arrive, 0, sizeof(int32_t)));
    HIP_TRY(hipDeviceSynchronize());  // the memset ran on the null stream; later work uses a non-blocking one
    *out = c;
    return 0;
}

void cafehip_destroy(cafehip_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    disarm(c);
    (void)sync_streams(c);
    for (auto& e : c->mc.e) hipEventDestroy(e.ready);
    if (c->mc.chain_end) hipEventDestroy(c->mc.chain_end);
    if (c->mc.stream) hipStreamDestroy(c->mc.stream);
    free_family_buffers(c);
    free_compression(c);
    delete c->link;
    hipFree(c->d_packed);
    hipFree(c->d_gathered);
    if (c->ev_x0) hipEventDestroy(c->ev_x0);
    if (c->ev_x1) hipEventDestroy(c->ev_x1);
    hipFree(c->d_ops);
    hipFree(c->d_mops);
    hipFree(c->d_park);
    hipFree(c->d_park_flags);
    hipFree(c->d_gen_done);
    hipFree(c->d_parent);
    hipFree(c->d_prefix);
    hipFree(c->d_vit_slot);
    hipFree(c->d_lncA);
    hipFree(c->d_lncB);
    hipFree(c->d_expA);
    hipFree(c->d_expB);
    if (c->h_fetch) hipHostFree(c->h_fetch);
    hipFree(c->d_vit);
    hipFree(c->d_PTfold);
    hipFree(c->d_PT);
    hipFree(c->d_node_key);
    hipFree(c->d_prior);
    hipFree(c->d_logprior);
    hipFree(c->d_err);
    hipFree(c->d_leaf_has_err);
    hipFree(c->d_leaf_has_err32);
    hipFree(c->d_first_zero);
    for (int i = 0; i < kParamRing; ++i) {
        if (c->h_params[i]) hipHostFree(c->h_params[i]);
        hipEventDestroy(c->h_params_ev[i]);
    }
    for (int i = 0; i < 4; ++i) hipEventDestroy(c->ev[i]);
    if (c->ev_mid) hipEventDestroy(c->ev_mid);
    hipHostFree(c->h_result);
    if (c->h_gate) hipHostFree(c->h_gate);
    hipFree(c->d_arrive);
    hipStreamDestroy(c->own_stream);
    delete c;
}

int cafehip_set_option(cafehip_ctx* c, const char* key, const char* value)
{
    if (!c || !key) return fail("null argument");
    disarm(c);   // (a pre-armed chain was queued under the old switches)
    return set_option(c, key, value ? value : "");
}

int cafehip_get_option(cafehip_ctx* c, const char* key, char* value, size_t value_bytes)
{
    if (!c || !key || !value || value_bytes == 0) return fail("null argument");
    const std::string k = key;
    const auto& o = c->opt;
    std::string v;
    if (k == "k2") v = o.k2 == 1 ? "v1" : (o.k2 == 2 ? "v1ref" : "auto");
    else if (k == "k1") v = o.k1 == 1 ? "exact" : (o.k1 == 2 ? "perterm" : "auto");
    else if (k == "compress") v = std::to_string(o.compress);
    else if (k == "errfold") v = std::to_string(o.errfold);
    else if (k == "matrix_cache") v = std::to_string(c->mc.want_entries);
    else if (k == "prefetch_where") v = std::to_string(o.prefetch_where);
    else if (k == "comm") v = c->comm_mode == 1 ? "rccl" : (c->comm_mode == 2 ? "direct" : "auto");
    else return fail("option '%s' cannot be read back", key);
    if (v.size() + 1 > value_bytes) return fail("option value needs %zu bytes", v.size() + 1);
    memcpy(value, v.c_str(), v.size() + 1);
    return 0;
}

int cafehip_set_stream(cafehip_ctx* c, void* hip_stream)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // NULL is a real stream in HIP (the legacy default stream, which is also what
    // torch.cuda.current_stream().cuda_stream returns unless a side stream is current)
    c->stream = (hipStream_t)hip_stream;
    return 0;
}

int cafehip_get_stream(cafehip_ctx* c, void** hip_stream)
{
    if (!c || !hip_stream) return fail("null argument");
    disarm(c);   // (a pre-armed chain waits on this stream)
    *hip_stream = (void*)c->stream;
    return 0;
}

int cafehip_set_tree(cafehip_ctx* c, int n_nodes, const int32_t* parent, const int32_t* left,
                     const int32_t* right, const double* branchlength)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    c->tune.n_items = -1;  // a new problem: measure the wave grids again
    if (n_nodes < 3 || (n_nodes & 1) == 0) return fail("a binary tree has an odd number (>= 3) of nodes, got %d", n_nodes);
    if (n_nodes > kMaxNodesCap) return fail("at most %d nodes supported, got %d", kMaxNodesCap, n_nodes);
    HIP_TRY(hipSetDevice(c->device));
    int root = -1;
    for (int i = 0; i < n_nodes; ++i) {
        const bool leaf = left[i] < 0;
        if (leaf != (right[i] < 0)) return fail("node %d has exactly one child: tree must be binary", i);
        if (leaf != ((i & 1) == 0))
            return fail("node %d: even ids must be leaves and odd ids internal (nlist in-order numbering)", i);
        if (!leaf && (left[i] >= n_nodes || right[i] >= n_nodes)) return fail("node %d: child out of range", i);
        if (parent[i] < 0) {
            if (root >= 0) return fail("two roots (%d and %d)", root, i);
            root = i;
        } else if (parent[i] >= n_nodes || (left[parent[i]] != i && right[parent[i]] != i)) {
            return fail("node %d: parent %d does not list it as a child", i, parent[i]);
        }
    }
    if (root < 0 || left[root] < 0) return fail("no internal root node");
    c->n_nodes = n_nodes;
    c->root = root;
    c->parent.assign(parent, parent + n_nodes);
    c->left.assign(left, left + n_nodes);
    c->right.assign(right, right + n_nodes);
    c->bl.assign(branchlength, branchlength + n_nodes);
    c->bl_int.resize(n_nodes);
    for (int i = 0; i < n_nodes; ++i) c->bl_int[i] = (int)branchlength[i];  // cafe/cafe_tree.c:376
    c->sched = cafehip::build_schedule(n_nodes, root, c->left, c->right);
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(c->d_ops);
    c->d_ops = nullptr;
    HIP_TRY(hipMalloc(&c->d_ops, c->sched.ops.size() * sizeof(cafehip::PruneOp)));
    HIP_TRY(hipMemcpy(c->d_ops, c->sched.ops.data(), c->sched.ops.size() * sizeof(cafehip::PruneOp),
                      hipMemcpyHostToDevice));
    c->msched = cafehip::build_mfma_schedule(n_nodes, root, c->left, c->right);
    hipFree(c->d_mops);
    c->d_mops = nullptr;
    HIP_TRY(hipMalloc(&c->d_mops, c->msched.ops.size() * sizeof(cafehip::MfmaOp)));
    HIP_TRY(hipMemcpy(c->d_mops, c->msched.ops.data(), c->msched.ops.size() * sizeof(cafehip::MfmaOp),
                      hipMemcpyHostToDevice));
    {
        // prefix order (tree_traveral_prefix, libtree/tree.c:101-124) and argmax-table slots for K4
        std::vector<int32_t> prefix, vslot(n_nodes, -1);
        std::vector<int> st;
        st.push_back(root);
        while (!st.empty()) {
            const int v = st.back();
            st.pop_back();
            prefix.push_back(v);
            if (c->left[v] >= 0) {
                st.push_back(c->right[v]);
                st.push_back(c->left[v]);
            }
        }
        int nt = 0;
        for (int i = 0; i < n_nodes; ++i)
            if (c->left[i] >= 0 && i != root) vslot[i] = nt++;
        c->n_vit_tables = nt;
        hipFree(c->d_parent);
        hipFree(c->d_prefix);
        hipFree(c->d_vit_slot);
        c->d_parent = c->d_prefix = c->d_vit_slot = nullptr;
        HIP_TRY(hipMalloc(&c->d_parent, n_nodes * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&c->d_prefix, n_nodes * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&c->d_vit_slot, n_nodes * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(c->d_parent, c->parent.data(), n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_prefix, prefix.data(), n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_vit_slot, vslot.data(), n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    c->have_matrices = false;
    if (mc_drop(c)) return -1;   // (entries are sized by the tree)
    if (ensure_param_ring(c)) return -1;
    if (c->M >= 0 && mc_prepare(c) < 0) return -1;
    if (c->M >= 0 && ensure_matrix_storage(c)) return -1;
    if (c->M >= 0 && rebuild_compression(c)) return -1;
    return 0;
}

int cafehip_set_families(cafehip_ctx* c, int F, int n_leaves, const int32_t* counts,
                         const int32_t* ref, int range_min, int range_max, int root_min,
                         int root_max)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    c->tune.n_items = -1;  // a new problem: measure the wave grids again
    if (F < 0 || n_leaves <= 0 || n_leaves > (kMaxNodesCap + 1) / 2) return fail("bad table shape %d x %d", F, n_leaves);
    if (range_min != 0) return fail("range_min must be 0 (cafe/cafe_family.c:357-364), got %d", range_min);
    if (range_max < 0 || root_min < 0 || root_max < root_min) return fail("bad ranges");
    const int R = root_max - root_min + 1;
    if (R > kMaxPrior) return fail("root range %d exceeds FAMILYSIZEMAX %d", R, kMaxPrior);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const auto t_setup0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    // CAFEHIP_SETUP_LOG=1: where the set-up time goes, lap by lap, on stderr (tools/setup_laps.py)
    static const bool lap_log = getenv("CAFEHIP_SETUP_LOG") != nullptr;
    auto t_lap = t_setup0;
    auto lap = [&](const char* what) {
        if (!lap_log) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "set_families lap %-28s %8.3f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_lap).count());
        t_lap = now;
    };
    // the reference asserts 0 <= familysize < size_of_factor (cafe/cafe_tree.c:207)
    const int sof = std::max(R, range_max + 1);
    for (size_t i = 0; i < (size_t)F * n_leaves; ++i)
        if (counts[i] < 0 || counts[i] >= sof)
            return fail("count %d at row %zu col %zu outside [0,%d)", counts[i], i / n_leaves, i % n_leaves, sof);

    // duplicate rows: ref = lowest identical index (cafe/cafe_family.c:9-34), hashed
    std::vector<int32_t> uniq_rows;
    c->fam2u.assign(F, 0);
    {
        // open-addressing table of row indices keyed by a 64-bit mix of the row (rows compared in full on a hit):
        // 100 k rows of 32 counts in ~2 ms (a map of std::string keys took 13)
        size_t cap = 16;
        while (cap < (size_t)F * 2) cap <<= 1;
        std::vector<int32_t> slot(ref ? 0 : cap, -1);
        const size_t row_bytes = sizeof(int32_t) * n_leaves;
        auto row_hash = [&](const int32_t* r) {
            uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n_leaves;
            for (int j = 0; j < n_leaves; ++j) {
                h ^= (uint32_t)r[j];
                h *= 0xD6E8FEB86659FD93ull;
                h ^= h >> 32;
            }
            return h;
        };
        for (int i = 0; i < F; ++i) {
            int rep;
            if (ref) {
                rep = (ref[i] < 0) ? i : ref[i];
                if (rep > i || rep < 0) return fail("ref[%d] = %d is not a lower-or-equal index", i, ref[i]);
                if (rep != i && memcmp(counts + (size_t)rep * n_leaves, counts + (size_t)i * n_leaves,
                                       sizeof(int32_t) * n_leaves) != 0)
                    return fail("ref[%d] = %d points at a different row", i, ref[i]);
            } else {
                const int32_t* row = counts + (size_t)i * n_leaves;
                size_t at = (size_t)row_hash(row) & (cap - 1);
                rep = i;
                while (slot[at] >= 0) {
                    if (memcmp(counts + (size_t)slot[at] * n_leaves, row, row_bytes) == 0) {
                        rep = slot[at];
                        break;
                    }
                    at = (at + 1) & (cap - 1);
                }
                if (rep == i) slot[at] = i;
            }
            if (rep == i) {
                c->fam2u[i] = (int32_t)uniq_rows.size();
                uniq_rows.push_back(i);
            } else {
                c->fam2u[i] = c->fam2u[rep];
            }
        }
    }
    const int Fu = (int)uniq_rows.size();
    c->setup_ms[0] = ms_since(t_setup0);
    lap("validate + dedup");
    const auto t_setup1 = std::chrono::steady_clock::now();
    std::vector<int32_t> ucounts((size_t)std::max(Fu, 1) * n_leaves, 0);
    for (int u = 0; u < Fu; ++u)
        memcpy(&ucounts[(size_t)u * n_leaves], counts + (size_t)uniq_rows[u] * n_leaves, sizeof(int32_t) * n_leaves);

    lap("unique rows gathered");
    free_family_buffers(c);
    lap("free old buffers");
    c->h_ucounts = ucounts;
    if (Fu == 0) c->h_ucounts.clear();
    c->out_sets = 1;
    c->F = F;
    c->Fu = Fu;
    c->n_leaves = n_leaves;
    c->range_min = range_min;
    c->range_max = range_max;
    c->root_min = root_min;
    c->root_max = root_max;
    const int M = std::max(range_max, root_max);  // cafe/cafe_main.c:325
    const bool new_M = (M != c->M);
    c->M = M;
    c->S = M + 1;
    c->C = range_max - range_min + 1;
    c->R = R;
    c->KP = ((c->S + 3) / 4) * 4;
    c->LD = ((c->S + 15) / 16) * 16 + 16;  // room for the root row offset + a 16-row tile overrun
    {
        // node-vector stride: the MFMA kernel stores whole 16-row tiles, so it must cover
        // roundup16(max(C, R)); congruent 2 mod 32 (conflict-free 16-family x 2-k LDS reads)
        const int need = ((std::max(c->C, c->R) + 15) / 16) * 16;
        c->LDv = 32 * ((std::max(need - 2, 0) + 31) / 32) + 2;
    }
    c->n_chunks = (F + CAFEHIP_CHUNK - 1) / CAFEHIP_CHUNK;

    HIP_TRY(hipMalloc(&c->d_counts, std::max<size_t>(ucounts.size(), 1) * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(c->d_counts, ucounts.data(), ucounts.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_fam2u, std::max(F, 1) * sizeof(int32_t)));
    if (F) HIP_TRY(hipMemcpy(c->d_fam2u, c->fam2u.data(), F * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_max_lik, std::max(Fu, 1) * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_max_post, std::max(Fu, 1) * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_argmax, std::max(Fu, 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->d_chunk_sums, std::max(c->n_chunks, 1) * sizeof(double)));
    if (!c->h_result || (size_t)c->n_chunks > c->h_result_chunks) {
        hipHostFree(c->h_result);
        c->h_result = nullptr;
        const size_t bytes = sizeof(HostResult) + (size_t)std::max(c->n_chunks, 1) * sizeof(double);
        HIP_TRY(hipHostMalloc((void**)&c->h_result, bytes, hipHostMallocMapped | hipHostMallocCoherent));
        memset((void*)c->h_result, 0, bytes);
        c->h_result_chunks = c->n_chunks;
    }
    lap("table + output buffers");
    if (new_M) {
        if (mc_drop(c)) return -1;   // (entries are sized by the matrix side; their builds read the tables released below)
        c->lnc.build(M);
        lap("ln C tables (host)");
        hipFree(c->d_lncA);
        hipFree(c->d_lncB);
        c->d_lncA = c->d_lncB = nullptr;
        HIP_TRY(hipMalloc(&c->d_lncA, c->lnc.A.size() * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_lncB, c->lnc.B.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(c->d_lncA, c->lnc.A.data(), c->lnc.A.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_lncB, c->lnc.B.data(), c->lnc.B.size() * sizeof(double), hipMemcpyHostToDevice));
        hipFree(c->d_expA);
        hipFree(c->d_expB);
        c->d_expA = c->d_expB = nullptr;
        HIP_TRY(hipMalloc(&c->d_expA, c->lnc.EA.size() * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_expB, c->lnc.EB.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(c->d_expA, c->lnc.EA.data(), c->lnc.EA.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_expB, c->lnc.EB.data(), c->lnc.EB.size() * sizeof(double), hipMemcpyHostToDevice));
        hipFree(c->d_PT);
        c->d_PT = nullptr;
        c->pt_keys_cap = 0;
        c->mc.slots_allocated = 0;
        c->have_matrices = false;
    }
    mc_invalidate(c);   // (a new table keeps the ranges' matrices valid in principle; the host driver starts a new search anyway)
    lap("ln C upload");
    if (c->n_nodes > 0 && mc_prepare(c) < 0) return -1;
    if (c->n_nodes > 0 && ensure_matrix_storage(c)) return -1;
    lap("matrix storage");
    c->setup_ms[2] = ms_since(t_setup1);
    const auto t_setup2 = std::chrono::steady_clock::now();
    if (rebuild_compression(c)) return -1;
    lap("compression plan + upload");
    c->setup_ms[1] = ms_since(t_setup2);
    c->setup_ms[3] = ms_since(t_setup0);
    return 0;
}

int cafehip_last_setup_ms(cafehip_ctx* c, double ms[4])
{
    if (!c || !ms) return fail("null argument");
    for (int i = 0; i < 4; ++i) ms[i] = c->setup_ms[i];
    return 0;
}

int cafehip_set_error_model(cafehip_ctx* c, int mfs, const double* errormatrix,
                            const uint8_t* leaf_has_model)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    c->tune.n_items = -1;  // a new problem: measure the wave grids again
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (mc_drop(c)) return -1;   // (matrices built ahead of time carry the old model's fold, or none)
    hipFree(c->d_err);
    hipFree(c->d_leaf_has_err);
    hipFree(c->d_leaf_has_err32);
    c->d_err = nullptr;
    c->d_leaf_has_err = nullptr;
    c->d_leaf_has_err32 = nullptr;
    c->err_mfs = -1;
    c->h_leaf_has_err.clear();
    if (!errormatrix) return upload_col_has_err(c) ? -1 : (mc_prepare(c) < 0 ? -1 : 0);
    if (c->n_nodes <= 0) return fail("set the tree before the error model");
    if (mfs < 0) return fail("bad error-model size %d", mfs);
    const size_t n = (size_t)(mfs + 1) * (mfs + 1);
    {
        // band of the model: non-zeros only where dlo <= true - observed <= dhi
        int dlo = INT_MAX, dhi = INT_MIN;
        for (int o = 0; o <= mfs; ++o)
            for (int t = 0; t <= mfs; ++t)
                if (errormatrix[(size_t)o * (mfs + 1) + t] != 0.0) {
                    dlo = std::min(dlo, t - o);
                    dhi = std::max(dhi, t - o);
                }
        if (dlo > dhi) dlo = dhi = 0;
        c->err_dlo = dlo;
        c->err_dhi = dhi;
        c->err_band_width = dhi - dlo + 1;
        c->err_banded = c->err_band_width <= 16 && c->opt.errband;
    }
    HIP_TRY(hipMalloc(&c->d_err, n * sizeof(double)));
    HIP_TRY(hipMemcpy(c->d_err, errormatrix, n * sizeof(double), hipMemcpyHostToDevice));
    std::vector<uint8_t> by_col((c->n_nodes + 1) / 2, 0);
    for (int i = 0; i < c->n_nodes; i += 2) by_col[i / 2] = leaf_has_model ? leaf_has_model[i] : 1;
    HIP_TRY(hipMalloc(&c->d_leaf_has_err, by_col.size()));
    HIP_TRY(hipMemcpy(c->d_leaf_has_err, by_col.data(), by_col.size(), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> w(by_col.begin(), by_col.end());
        HIP_TRY(hipMalloc(&c->d_leaf_has_err32, w.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(c->d_leaf_has_err32, w.data(), w.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    c->h_leaf_has_err = by_col;
    c->err_mfs = mfs;
    if (upload_col_has_err(c)) return -1;
    return mc_prepare(c) < 0 ? -1 : 0;   // (the entries again, with room for the folded twins)
}

int cafehip_num_chunks(cafehip_ctx* c) { return c ? c->n_chunks : fail("null context"); }
int cafehip_matrix_size(cafehip_ctx* c) { return c ? c->S : fail("null context"); }

int cafehip_eval_posterior_async(cafehip_ctx* c, const double* node_lambda, const double* node_mu,
                                 const double* prior, double* d_chunk_sums, int32_t* d_first_zero)
{
    if (!c) return fail("null context");
    if (!node_lambda || !node_mu || !prior || !d_chunk_sums || !d_first_zero) return fail("null argument");
    if (c->d_err && c->err_mfs < c->range_max)
        return fail("error model covers sizes 0..%d but range_max is %d", c->err_mfs, c->range_max);
    return eval_device(c, node_lambda, node_mu, prior, d_chunk_sums, d_first_zero);
}

int cafehip_eval_posterior(cafehip_ctx* c, const double* node_lambda, const double* node_mu,
                           const double* prior, double* score, int32_t* first_zero_family,
                           double* max_lik, int32_t* argmax_root, double* max_post)
{
    if (!c) return fail("null context");
    if (!node_lambda || !node_mu || !prior || !score) return fail("null argument");
    if (c->d_err && c->err_mfs < c->range_max)
        return fail("error model covers sizes 0..%d but range_max is %d", c->err_mfs, c->range_max);
    if (eval_device(c, node_lambda, node_mu, prior, c->d_chunk_sums, c->d_first_zero, true)) return -1;
    if (c->n_chunks > 0) {
        // spin on the sequence number the last K3 block publishes (a few microseconds after the kernel
        // ends); fall back to a stream query now and then so that a faulted launch cannot hang us
        const int32_t want = c->eval_seq;
        unsigned long spins = 0;
        while (c->h_result->done_seq != want) {
            if ((++spins & 0x3FFFF) == 0) {
                hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) {
                    if (c->h_result->done_seq != want) HIP_TRY(hipStreamSynchronize(c->stream));
                    if (c->h_result->done_seq != want) return fail("score kernel finished without publishing its result");
                    break;
                }
                if (q != hipErrorNotReady) return fail("stream error while waiting: %s", hipGetErrorString(q));
            }
        }
    } else {
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    // the payload below was written before the sequence number (device-side system fence): order our reads after
    // the flag read
    std::atomic_thread_fence(std::memory_order_acquire);
    if (armed_chain_expired(c)) {
        // (the gate gave up in the instant the parameters were being staged: the chain ran on a block in flux)
        const int keep = c->opt.prearm;
        c->opt.prearm = 0;
        disarm(c);
        const int rc = cafehip_eval_posterior(c, node_lambda, node_mu, prior, score, first_zero_family, max_lik, argmax_root, max_post);
        c->opt.prearm = keep;
        return rc;
    }
    if (collect_kernel_ms(c)) return -1;
    // fixed-order final sum over chunks (independent of how chunks were produced)
    double s = 0.0;
    for (int i = 0; i < c->n_chunks; ++i) s += c->h_result->chunk_sums[i];
    const int32_t hfz = c->n_chunks > 0 ? c->h_result->first_zero[0] : INT32_MAX;
    const int fz = (hfz >= 0 && hfz < c->F) ? hfz : -1;
    *score = (fz >= 0) ? -INFINITY : s;  // cafe/lambda.cpp:753-760
    if (first_zero_family) *first_zero_family = fz;
    if (max_lik || argmax_root || max_post) {
        std::vector<double> ml(c->Fu), mp(c->Fu);
        std::vector<int32_t> am(c->Fu);
        if (c->Fu) {
            HIP_TRY(hipMemcpy(ml.data(), c->d_max_lik, c->Fu * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(mp.data(), c->d_max_post, c->Fu * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(am.data(), c->d_argmax, c->Fu * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < c->F; ++i) {
            const int u = c->fam2u[i];
            if (max_lik) max_lik[i] = ml[u];
            if (max_post) max_post[i] = mp[u];
            if (argmax_root) argmax_root[i] = am[u];
        }
    }
    return 0;
}

int cafehip_eval_posterior_multi(cafehip_ctx* c, int n_sets, const double* node_lambda, const double* node_mu,
                                 const double* prior, double* scores, int32_t* first_zero_family)
{
    if (!c) return fail("null context");
    if (!node_lambda || !node_mu || !prior || !scores) return fail("null argument");
    if (n_sets < 1 || n_sets > kMaxSets) return fail("1..%d parameter sets per call, got %d", kMaxSets, n_sets);
    if (c->d_err && c->err_mfs < c->range_max)
        return fail("error model covers sizes 0..%d but range_max is %d", c->err_mfs, c->range_max);
    if (n_sets == 1) return cafehip_eval_posterior(c, node_lambda, node_mu, prior, scores, first_zero_family, nullptr, nullptr, nullptr);
    if (eval_device(c, node_lambda, node_mu, prior, nullptr, c->d_first_zero, true, n_sets)) return -1;
    if (c->n_chunks > 0) {
        const int32_t want = c->eval_seq;
        unsigned long spins = 0;
        while (c->h_result->done_seq != want) {
            if ((++spins & 0x3FFFF) == 0) {
                hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) {
                    if (c->h_result->done_seq != want) HIP_TRY(hipStreamSynchronize(c->stream));
                    if (c->h_result->done_seq != want) return fail("score kernel finished without publishing its result");
                    break;
                }
                if (q != hipErrorNotReady) return fail("stream error while waiting: %s", hipGetErrorString(q));
            }
        }
    } else {
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (collect_kernel_ms(c)) return -1;
    for (int q = 0; q < n_sets; ++q) {
        // the same fixed-order sum over this set's chunks as the single-set call: bit-identical scores
        double sum = 0.0;
        for (int i = 0; i < c->n_chunks; ++i) sum += c->h_result->chunk_sums[(size_t)q * c->n_chunks + i];
        const int32_t hfz = c->n_chunks > 0 ? c->h_result->first_zero[q] : INT32_MAX;
        const int fz = (hfz >= 0 && hfz < c->F) ? hfz : -1;
        scores[q] = (fz >= 0) ? -INFINITY : sum;
        if (first_zero_family) first_zero_family[q] = fz;
    }
    return 0;
}

int cafehip_eval_posterior_sequence(cafehip_ctx* c, int n, const double* node_lambda, const double* node_mu, const double* prior,
                                    double* scores, int32_t* first_zero_family, int sharded)
{
    if (!c) return fail("null context");
    if (n < 0 || (n > 0 && (!node_lambda || !node_mu || !prior || !scores))) return fail("bad argument");
    // one evaluation after the other, each complete (its score on the host) before the next is staged: exactly the calls a
    // caller's own loop would make, without the caller's per-call overhead
    for (int i = 0; i < n; ++i) {
        const double* nl = node_lambda + (size_t)i * c->n_nodes;
        const double* nm = node_mu + (size_t)i * c->n_nodes;
        int32_t fz = -1;
        const int rc = sharded ? cafehip_eval_posterior_sharded(c, nl, nm, prior, scores + i, &fz)
                               : cafehip_eval_posterior(c, nl, nm, prior, scores + i, &fz, nullptr, nullptr, nullptr);
        if (rc != 0) return -1;
        if (first_zero_family) first_zero_family[i] = fz;
    }
    return 0;
}

int cafehip_eval_clustered_posterior(cafehip_ctx* c, int K, const double* node_lambda, const double* node_mu,
                                     const double* weights, const double* prior, double* score,
                                     int32_t* first_zero_family, double* membership_sums, double* family_map,
                                     double* family_membership)
{
    if (check_ready(c)) return -1;
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (!node_lambda || !node_mu || !weights || !prior || !score || !membership_sums) return fail("null argument");
    if (K < 1 || K > kMaxSets) return fail("1..%d clusters, got %d", kMaxSets, K);
    if (c->d_err && c->err_mfs < c->range_max)
        return fail("error model covers sizes 0..%d but range_max is %d", c->err_mfs, c->range_max);
    HIP_TRY(hipSetDevice(c->device));
    if (ensure_output_sets(c, std::max(K, 2))) return -1;
    if (stage_params(c, node_lambda, node_mu, prior, K)) return -1;
    c->fz_clean = false;   // (k3_cluster_score leaves the first-zero word as it found it)
    if (launch_k1(c, c->d_first_zero)) return -1;
    if (launch_error_fold(c)) return -1;
    K2Args a;
    fill_common_k2(c, a);
    a.counts = c->d_counts;
    a.Fu = c->Fu;
    a.max_lik = c->d_max_lik;
    a.argmax = c->d_argmax;
    a.max_post = c->d_max_post;
    if (c->d_err) {
        a.err = c->d_err;
        a.err_ld = c->err_mfs + 1;
        a.leaf_has_err = c->d_leaf_has_err;
    }
    if (K == 1) {
        if (launch_k2(c, a, c->Fu, 1)) return -1;
    } else if (launch_k2(c, a, c->Fu, K)) {
        return -1;
    }
    std::fill(membership_sums, membership_sums + K, 0.0);
    *score = 0.0;
    if (first_zero_family) *first_zero_family = -1;
    if (c->n_chunks == 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }
    ClusterWeights cw;
    for (int k = 0; k < kMaxSets; ++k) cw.w[k] = k < K ? weights[k] : 0.0;
    double *d_memb = nullptr, *d_map = nullptr, *d_pz = nullptr;
    auto cleanup = [&]() { hipFree(d_memb); hipFree(d_map); hipFree(d_pz); };
#define TRY4(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    TRY4(hipMalloc(&d_memb, (size_t)K * c->n_chunks * sizeof(double)));
    if (family_map) TRY4(hipMalloc(&d_map, (size_t)c->F * sizeof(double)));
    if (family_membership) TRY4(hipMalloc(&d_pz, (size_t)c->F * K * sizeof(double)));
    {
        K3cArgs ka{c->d_max_post, c->d_fam2u, c->F, c->Fu, K, cw, c->d_chunk_sums, d_memb, c->d_first_zero, d_map, d_pz};
        if (launch_kernel(k3_cluster_kernel(), dim3(c->n_chunks), dim3(CAFEHIP_CHUNK), 0, c->stream, ka)) { cleanup(); return -1; }
    }
    std::vector<double> sums(c->n_chunks), memb((size_t)K * c->n_chunks);
    int32_t fz_dev = INT32_MAX;
    TRY4(hipMemcpyAsync(sums.data(), c->d_chunk_sums, sums.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY4(hipMemcpyAsync(memb.data(), d_memb, memb.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY4(hipMemcpyAsync(&fz_dev, c->d_first_zero, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (family_map) TRY4(hipMemcpyAsync(family_map, d_map, (size_t)c->F * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (family_membership)
        TRY4(hipMemcpyAsync(family_membership, d_pz, (size_t)c->F * K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY4(hipStreamSynchronize(c->stream));
#undef TRY4
    cleanup();
    double sc = 0.0;
    for (int i = 0; i < c->n_chunks; ++i) sc += sums[i];
    for (int k = 0; k < K; ++k) {
        double m = 0.0;
        for (int i = 0; i < c->n_chunks; ++i) m += memb[(size_t)k * c->n_chunks + i];
        membership_sums[k] = m;
    }
    const int fz = (fz_dev >= 0 && fz_dev < c->F) ? fz_dev : -1;
    *score = fz >= 0 ? -INFINITY : sc;
    if (first_zero_family) *first_zero_family = fz;
    return 0;
}

int cafehip_launch_info(cafehip_ctx* c, int* k2_workgroups, int* compute_units)
{
    if (!c) return fail("null context");
    if (k2_workgroups) *k2_workgroups = c->k2_used_mfma ? c->k2_grid : (c->k2_nf > 0 ? (c->Fu + c->k2_nf - 1) / c->k2_nf : 0);
    if (compute_units) *compute_units = c->n_cu;
    return 0;
}

int cafehip_last_issued_flops(cafehip_ctx* c, double* walk, double* tables)
{
    if (!c) return fail("null context");
    if (walk) *walk = c->issued_walk;
    if (tables) *tables = c->last_compressed ? c->issued_tables : 0.0;
    return 0;
}

int cafehip_prefetch_matrices(cafehip_ctx* c, int n_sets, const double* node_lambda, const double* node_mu, int when)
{
    if (check_ready(c)) return -1;
    if (n_sets < 0 || n_sets > kMaxSets) return fail("0..%d parameter sets per prefetch, got %d", kMaxSets, n_sets);
    if (n_sets > 0 && (!node_lambda || !node_mu)) return fail("null argument");
    if (when != CAFEHIP_PREFETCH_NOW && when != CAFEHIP_PREFETCH_BEHIND_NEXT_EVALUATION) return fail("prefetch: unknown `when` %d", when);
    auto& mc = c->mc;
    if (mc.want_entries <= 0 || mc.broken) return 0;   // a hint: ignored when the store is off
    HIP_TRY(hipSetDevice(c->device));
    if (when == CAFEHIP_PREFETCH_BEHIND_NEXT_EVALUATION) {
        mc.pending_sets = n_sets;   // (replaces an earlier request that no evaluation picked up)
        mc.pending_l.assign(node_lambda, node_lambda + (size_t)n_sets * c->n_nodes);
        mc.pending_m.assign(node_mu, node_mu + (size_t)n_sets * c->n_nodes);
        return 0;
    }
    mc.pending_sets = 0;
    return mc_build(c, n_sets, node_lambda, node_mu);
}

int cafehip_prearm_stats(cafehip_ctx* c, long out[3])
{
    if (!c || !out) return fail("null argument");
    out[0] = c->prearm_used;
    out[1] = c->prearm_wasted;
    out[2] = c->prearm_expired;
    return 0;
}

int cafehip_matrix_cache_stats(cafehip_ctx* c, long out[CAFEHIP_MATRIX_CACHE_STATS])
{
    if (!c || !out) return fail("null argument");
    const auto& mc = c->mc;
    out[0] = mc.requested;
    out[1] = mc.built;
    out[2] = mc.hits;
    out[3] = mc.misses;
    out[4] = mc.evicted;
    out[5] = mc.waited;
    out[6] = mc.launches;
    out[7] = (long)mc.e.size();
    return 0;
}

int cafehip_reset_birthdeath_cache(cafehip_ctx* c, const double* node_lambda, const double* node_mu)
{
    if (check_ready(c)) return -1;
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (!node_lambda || !node_mu) return fail("null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (stage_params(c, node_lambda, node_mu, nullptr)) return -1;
    if (launch_k1(c)) return -1;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int cafehip_set_exact_matrices(cafehip_ctx* c, int on)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (c->force_exact != (on != 0)) mc_invalidate(c);   // (matrices built ahead of time carry the other form)
    c->force_exact = on != 0;
    return 0;
}

int cafehip_get_matrix(cafehip_ctx* c, int node, double* out, int* S_out)
{
    if (check_ready(c)) return -1;
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (!c->have_matrices) return fail("no matrices built yet");
    if (node < 0 || node >= c->n_nodes || c->node_key[node] < 0) return fail("node %d has no matrix", node);
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->KP * c->LD;
    std::vector<double> pt(n);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(pt.data(), c->d_PT + (size_t)c->node_key[node] * n, n * sizeof(double), hipMemcpyDeviceToHost));
    for (int s = 0; s < c->S; ++s)
        for (int k = 0; k < c->S; ++k) out[(size_t)s * c->S + k] = pt[(size_t)k * c->LD + s];
    if (S_out) *S_out = c->S;
    return 0;
}

int cafehip_eval_root_likelihoods(cafehip_ctx* c, int B, const int32_t* counts, const int32_t* root_lo,
                                  const int32_t* root_hi, const int32_t* col_max, double* out)
{
    if (check_ready(c)) return -1;
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (!c->have_matrices) return fail("no matrices built yet (call cafehip_eval_posterior or cafehip_reset_birthdeath_cache)");
    if (B <= 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int64_t> off(B);
    int64_t total = 0;
    for (int b = 0; b < B; ++b) {
        if (root_lo[b] < c->root_min || root_hi[b] > c->root_max || root_hi[b] < root_lo[b])
            return fail("row %d: root range [%d,%d] outside [%d,%d]", b, root_lo[b], root_hi[b], c->root_min, c->root_max);
        if (col_max[b] < 0 || col_max[b] > c->range_max)
            return fail("row %d: col_max %d outside [0,%d]", b, col_max[b], c->range_max);
        for (int j = 0; j < c->n_leaves; ++j)
            if (counts[(size_t)b * c->n_leaves + j] < 0) return fail("row %d: negative count", b);
        off[b] = total;
        total += root_hi[b] - root_lo[b] + 1;
    }
    int32_t *d_cnt = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_cm = nullptr;
    int64_t* d_off = nullptr;
    double* d_out = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        hipFree(d_cnt); hipFree(d_lo); hipFree(d_hi); hipFree(d_cm); hipFree(d_off); hipFree(d_out);
    };
#define TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    TRY2(hipMalloc(&d_cnt, (size_t)B * c->n_leaves * sizeof(int32_t)));
    TRY2(hipMalloc(&d_lo, B * sizeof(int32_t)));
    TRY2(hipMalloc(&d_hi, B * sizeof(int32_t)));
    TRY2(hipMalloc(&d_cm, B * sizeof(int32_t)));
    TRY2(hipMalloc(&d_off, B * sizeof(int64_t)));
    TRY2(hipMalloc(&d_out, total * sizeof(double)));
    TRY2(hipMemcpyAsync(d_cnt, counts, (size_t)B * c->n_leaves * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY2(hipMemcpyAsync(d_lo, root_lo, B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY2(hipMemcpyAsync(d_hi, root_hi, B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY2(hipMemcpyAsync(d_cm, col_max, B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY2(hipMemcpyAsync(d_off, off.data(), B * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    K2Args a;
    fill_common_k2(c, a);
    a.counts = d_cnt;
    a.Fu = B;
    a.root_lo = d_lo;
    a.root_hi = d_hi;
    a.col_max = d_cm;
    a.out_off = d_off;
    a.out_root = d_out;
    // the reference drops the error model on tree copies (cafe/cafe_tree.c:485-494): not applied here
    double* saved_err = c->d_err;
    c->d_err = nullptr;
    if (c->timing) TRY2(hipEventRecord(c->ev[1], c->stream));
    rc = launch_k2(c, a, B);
    c->d_err = saved_err;
    if (rc) { cleanup(); return -1; }
    if (c->k2_used_mfma && c->opt.batch_trim && c->k2_nf > 0) {
        // matrix-instruction flops this launch issues with every tile trimmed to its largest column limit / its root sizes
        // (the same rule as the kernel's prologue)
        const int nf = c->k2_nf, ks_full = (c->C + 3) / 4;
        double issued = 0;
        for (int b0 = 0; b0 < B; b0 += nf) {
            int kmax = 0, rlo = INT_MAX, rhi = 0;
            for (int b = b0; b < std::min(B, b0 + nf); ++b) {
                kmax = std::max(kmax, (int)col_max[b]);
                rlo = std::min(rlo, root_lo[b] - c->root_min);
                rhi = std::max(rhi, root_hi[b] - c->root_min);
            }
            const int ks = std::min(ks_full, (kmax + 4) >> 2), rt = std::min((c->C + 15) / 16, (kmax + 16) >> 4);
            const int rt_root = std::min(rhi >> 4, (c->R + 15) / 16 - 1) + 1 - (std::min(rlo, rhi) >> 4);
            for (const auto& op : c->msched.ops)
                issued += 2.0 * 4.0 * ks * 16.0 * (op.is_root ? rt_root : rt) * ((op.kind[0] == 1) + (op.kind[1] == 1)) * nf;
        }
        c->issued_walk = issued;
    }
    if (c->timing) TRY2(hipEventRecord(c->ev[2], c->stream));
    TRY2(hipMemcpyAsync(out, d_out, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY2(hipStreamSynchronize(c->stream));
    if (c->timing) {
        float ms = 0;
        TRY2(hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
        c->last_batch_ms = ms;
    }
#undef TRY2
    cleanup();
    return 0;
}

int cafehip_viterbi(cafehip_ctx* c, int B, const int32_t* counts, const int32_t* root_lo,
                    const int32_t* root_hi, const int32_t* col_max, int32_t* node_sizes)
{
    if (check_ready(c)) return -1;
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (!c->have_matrices) return fail("no matrices built yet (call cafehip_eval_posterior or cafehip_reset_birthdeath_cache)");
    if (B <= 0) return 0;
    if (!counts || !root_lo || !root_hi || !col_max || !node_sizes) return fail("null argument");
    if (c->C > 65535) return fail("matrix side too large for 16-bit argmax tables");
    HIP_TRY(hipSetDevice(c->device));
    for (int b = 0; b < B; ++b) {
        if (root_lo[b] < c->root_min || root_hi[b] > c->root_max)
            return fail("row %d: root range [%d,%d] outside [%d,%d]", b, root_lo[b], root_hi[b], c->root_min, c->root_max);
        if (col_max[b] < 0 || col_max[b] > c->range_max) return fail("row %d: col_max %d outside [0,%d]", b, col_max[b], c->range_max);
        for (int j = 0; j < c->n_leaves; ++j)
            if (counts[(size_t)b * c->n_leaves + j] < 0) return fail("row %d: negative count", b);
    }
    const int rows_max = std::max(c->C, c->R);
    const int block = ((rows_max + 63) / 64) * 64;
    if (block > 1024) return fail("matrix side %d exceeds the 1024 rows this kernel handles", rows_max);
    int nf = 8;
    size_t lds = 0;
    const size_t stat = 64;   // the kernel's static LDS (column limits)
    const auto cnt_bytes = [&](int nf_) { return (size_t)((nf_ * c->n_leaves + 1) & ~1) * sizeof(int); };   // counts behind the slots
    // argmax tables in global scratch (default): LDS holds the node-vector slots only, sized for >= 4 workgroups
    // per CU; option vitlds=1 keeps the tables in LDS (one workgroup per CU at the larger shapes)
    bool tables_global = c->opt.vitlds != 1;
    const size_t per_family_tables = (size_t)c->n_vit_tables * c->LDv * sizeof(unsigned short);
    if (tables_global) {
        for (nf = 8; nf >= 1; nf >>= 1) {
            lds = (size_t)c->sched.n_slots * nf * c->LDv * sizeof(double) + cnt_bytes(nf) + 16;
            if (lds + stat <= (size_t)40 * 1024 || nf == 1) break;
        }
        if (lds + stat > (size_t)c->lds_limit) return fail("Viterbi node vectors of this tree do not fit LDS");
        const size_t max_batch = 65536;
        const size_t need = std::min<size_t>((size_t)B, max_batch) * per_family_tables + 8 * per_family_tables;
        if (need > c->vit_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            hipFree(c->d_vit);
            c->d_vit = nullptr;
            c->vit_cap = 0;
            if (hipMalloc(&c->d_vit, need) == hipSuccess) c->vit_cap = need;
            else { (void)hipGetLastError(); tables_global = false; }   // no room: tables in LDS as before
        }
    }
    if (!tables_global) {
        for (nf = 8; nf >= 1; nf >>= 1) {
            lds = (size_t)c->sched.n_slots * nf * c->LDv * sizeof(double) + cnt_bytes(nf) + (size_t)nf * per_family_tables + 16;
            if (lds + stat <= (size_t)c->lds_limit) break;
        }
        if (nf < 1) return fail("Viterbi tables of this tree do not fit LDS");
    }
    int32_t *d_cnt = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_cm = nullptr, *d_out = nullptr;
    auto cleanup = [&]() { hipFree(d_cnt); hipFree(d_lo); hipFree(d_hi); hipFree(d_cm); hipFree(d_out); };
#define TRY3(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    TRY3(hipMalloc(&d_cnt, (size_t)B * c->n_leaves * sizeof(int32_t)));
    TRY3(hipMalloc(&d_lo, B * sizeof(int32_t)));
    TRY3(hipMalloc(&d_hi, B * sizeof(int32_t)));
    TRY3(hipMalloc(&d_cm, B * sizeof(int32_t)));
    TRY3(hipMalloc(&d_out, (size_t)B * c->n_nodes * sizeof(int32_t)));
    TRY3(hipMemcpyAsync(d_cnt, counts, (size_t)B * c->n_leaves * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY3(hipMemcpyAsync(d_lo, root_lo, B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY3(hipMemcpyAsync(d_hi, root_hi, B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRY3(hipMemcpyAsync(d_cm, col_max, B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    K4Args a;
    memset(&a, 0, sizeof a);
    a.PT = c->d_PT;
    a.node_key = c->cur_node_key;
    a.ops = c->d_ops;
    a.n_ops = (int)c->sched.ops.size();
    a.counts = d_cnt;
    a.B = B;
    a.n_leaves = c->n_leaves;
    a.n_nodes = c->n_nodes;
    a.C = c->C;
    a.R = c->R;
    a.root_min = c->root_min;
    a.LD = c->LD;
    a.KP = c->KP;
    a.LDv = c->LDv;
    a.n_slots = c->sched.n_slots;
    a.root = c->root;
    a.parent = c->d_parent;
    a.prefix = c->d_prefix;
    a.vit_slot = c->d_vit_slot;
    a.n_tables = c->n_vit_tables;
    a.root_lo = d_lo;
    a.root_hi = d_hi;
    a.col_max = d_cm;
    a.node_sizes = d_out;
    int rc = 0;
    // sub-batches bound the table scratch (65,536 families = 1 GB at 31 internal nodes x 258 rows)
    const int batch = tables_global ? 65536 : B;
    for (int b0 = 0; b0 < B && rc == 0; b0 += batch) {
        K4Args ab = a;
        ab.B = std::min(batch, B - b0);
        ab.counts = d_cnt + (size_t)b0 * c->n_leaves;
        ab.root_lo = d_lo + b0;
        ab.root_hi = d_hi + b0;
        ab.col_max = d_cm + b0;
        ab.node_sizes = d_out + (size_t)b0 * c->n_nodes;
        ab.vit_global = tables_global ? c->d_vit : nullptr;
        rc = launch_k4_nf(c, nf, ab, block, lds);
    }
    if (rc) { cleanup(); return -1; }
    TRY3(hipMemcpyAsync(node_sizes, d_out, (size_t)B * c->n_nodes * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TRY3(hipStreamSynchronize(c->stream));
#undef TRY3
    cleanup();
    return 0;
}

int cafehip_fetch_small(cafehip_ctx* c, const void* d_src, size_t nbytes, const void** host_ptr)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    if (!d_src || !host_ptr || nbytes == 0 || (nbytes & 7) || nbytes > (1u << 20)) return fail("bad fetch of %zu bytes", nbytes);
    HIP_TRY(hipSetDevice(c->device));
    const size_t n_words = nbytes / 8;
    if (!c->h_fetch || c->h_fetch_words < n_words) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->h_fetch) hipHostFree(c->h_fetch);
        c->h_fetch = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->h_fetch, (n_words + 1) * 8, hipHostMallocMapped | hipHostMallocCoherent));
        c->h_fetch_words = n_words;
        c->h_fetch[0] = 0;
        c->fetch_seq = 0;
    }
    const int32_t want = ++c->fetch_seq;
    volatile int32_t* flag = reinterpret_cast<volatile int32_t*>(c->h_fetch);
    {
        FetchArgs fa{static_cast<const uint64_t*>(d_src), c->h_fetch + 1, n_words, flag, want};
        if (launch_kernel(fetch_small_kernel(), dim3(1), dim3(256), 0, c->stream, fa)) return -1;
    }
    unsigned long spins = 0;
    while (*flag != want) {
        if ((++spins & 0x3FFFF) == 0) {  // a faulted launch must not hang the caller
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (*flag != want) HIP_TRY(hipStreamSynchronize(c->stream));
                if (*flag != want) return fail("fetch kernel finished without publishing");
                break;
            }
            if (q != hipErrorNotReady) return fail("stream error while waiting: %s", hipGetErrorString(q));
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);  // payload reads stay behind the flag read
    *host_ptr = c->h_fetch + 1;
    return 0;
}

int cafehip_exp_like_host_selftest(long n, unsigned seed, long* mismatches_fused, long* mismatches_plain)
{
    // host only: both restated forms of exp() against this host's std::exp on n arguments; returns the variant K1's exact
    // form would use (1 fused, 2 plain, 0 neither)
    return host_exp_variant(mismatches_fused, mismatches_plain, n, seed);
}

int cafehip_enable_timing(cafehip_ctx* c, int on)
{
    if (!c) return fail("null context");
    disarm(c);   // (a pre-armed chain waits on this stream)
    c->timing = on != 0;
    return 0;
}

int cafehip_last_kernel_ms(cafehip_ctx* c, double ms[3])
{
    if (!c) return fail("null context");
    if (collect_kernel_ms(c)) return -1;  // the asynchronous entry point leaves the events pending
    for (int i = 0; i < 3; ++i) ms[i] = c->last_ms[i];
    return 0;
}

int cafehip_last_tables_ms(cafehip_ctx* c, double* ms)
{
    if (!c || !ms) return fail("null argument");
    if (collect_kernel_ms(c)) return -1;
    *ms = c->last_tables_ms;
    return 0;
}

int cafehip_last_batch_ms(cafehip_ctx* c, double* ms)
{
    if (!c || !ms) return fail("null argument");
    *ms = c->last_batch_ms;
    return 0;
}

const char* cafehip_describe(cafehip_ctx* c)
{
    if (!c) return "";
    char buf[512];
    snprintf(buf, sizeof buf,
             "device=%d cus=%d F=%d Fu=%d n_leaves=%d S=%d C=%d R=%d LD=%d KP=%d LDv=%d nkeys=%d "
             "n_ops=%zu n_slots=%d n_parks=%d k1:%s k2:%s NF=%d block=%d lds=%zu cfg(nftw,nrtw,wf,wr)=%d,%d,%d,%d grid=%d park_slots=%d",
             c->device, c->n_cu, c->F, c->Fu, c->n_leaves, c->S, c->C, c->R, c->LD, c->KP, c->LDv,
             c->nkeys, c->sched.ops.size(), c->sched.n_slots, c->msched.n_parks,
             c->k1_product_form ? "product" : "exact",
             c->k2_used_mfma ? (c->k2_shape4 ? "mfma4x4(cfg=G,nrtw,wf,wr)" : "mfma") : "v1", c->k2_nf, c->k2_block, c->k2_lds, c->k2_cfg[0], c->k2_cfg[1],
             c->k2_cfg[2], c->k2_cfg[3], c->k2_grid, c->k2_park_slots);
    c->desc = buf;
    if (c->cp.valid) {
        snprintf(buf, sizeof buf, " compressed(nodes=%d levels=%zu states=%ld walk_steps=%zu walk_cols=%d used=%d level_tiles=", c->cp.n_nodes,
                 c->cp.level_first.size() - 1, c->cp.states, c->cp.sched.ops.size(), c->cp.n_cols, (int)c->last_compressed);
        c->desc += buf;
        for (size_t l = 0; l + 1 < c->cp.level_first.size(); ++l) c->desc += (l ? "/" : "") + std::to_string(c->cp.level_first[l + 1] - c->cp.level_first[l]);
        c->desc += ")";
    }
    return c->desc.c_str();
}

}  // extern "C"
