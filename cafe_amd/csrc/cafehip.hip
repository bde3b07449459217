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

#include "matrix_store.hpp"

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


#include "compression_plan.hpp"

#include "k2_launch.hpp"

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
            // (the score blocks inherit K1's LDS and launch bounds: only where they are one round of the chip -- ADVICE r05)
            if (P.register_blocked && c->n_chunks <= std::max(c->n_cu, 1)) {
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
    // worth a microsecond per evaluation inside a steady loop (115.3 -> 114.3 us at configs[1], profiles/r05/prearm_ab.txt) and
    // only where an evaluation is short -- a chain let go unused repeats a whole evaluation.  Off by default: the END of a loop
    // pays (the next synchronisation of the stream waits out the gate's slice: bench.py's 100-step table legs measured
    // +0.2 ms per step = one 20 ms slice behind their closing barrier)
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
const char* const kOptionNames[] = {"compress", "compress_theta", "compress_min", "compress_max_level", "compress_drop_top", "errfold", "errband", "k1", "k1kpb", "k1_balance", "k2_small_r", "k2_objective_kernels", "k2_skip_epilogue", "k2", "mfma",
                                    "k2cfg", "k2cfg4", "k2tune", "k2tune_log", "k2slots", "ldspark", "vitlds", "k2c_batch", "k2c_pair", "k2c_pair_min", "k2c_gemm", "k2c_nst", "k2c_xcd",
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
    else if (key == "k2_small_r") o.k2_small_r = iv != 0;
    else if (key == "k2_objective_kernels") o.k2_objective_kernels = iv != 0;
    else if (key == "k1_balance") o.k1_balance = iv;   // 0 off, 1 by launch size, 2 / 3: forced, alternating / heaviest first; 11: forced halves
    else if (key == "k2_skip_epilogue") o.k2_skip_epilogue = iv != 0;
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
    else if (key == "k2c_pair") o.k2c_pair = val.empty() ? -1 : iv;
    else if (key == "k2c_pair_min") o.k2c_pair_min = std::max(iv, 0);
    else if (key == "k2c_gemm") { o.k2c_gemm = val.empty() ? -1 : iv; replan = true; }
    else if (key == "k2c_nst") {
        if (!(iv == 0 || iv == 1 || iv == 2 || iv == 4)) return fail("option k2c_nst: 0 (by level size) | 1 | 2 | 4, got '%s'", val.c_str());
        o.k2c_nst = iv;
        replan = true;
    }
    else if (key == "k2c_xcd") o.k2c_xcd = iv != 0;
    else if (key == "batch_trim") o.batch_trim = iv != 0;
    else if (key == "batch_lockstep") o.batch_lockstep = iv != 0;
    else if (key == "walk_lockstep") o.walk_lockstep = iv != 0;
    else if (key == "batch_lockstep_slack") o.batch_lockstep_slack = std::min(std::max(iv, 0), 100);
    else if (key == "exp_like_host") o.exp_like_host = iv != 0;
    else if (key == "prearm") { disarm(c); o.prearm = iv != 0; return 0; }
    else if (key == "test_stall_ms") { o.test_stall_ms = std::max(iv, 0); return 0; }   // test hook: the host sleeps before it looks for the score
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
namespace {
// The library's kernels live in one code object per translation unit, and the runtime loads a code object at the FIRST launch
// of one of its kernels: ~2.8 ms each for the two walk units (profiles/r06/cold_evaluations_before.txt: evaluation 1 of a
// process took 2.9 ms, the first candidate wave grid of the other matrix-instruction shape another 2.9 ms) -- 6 of the 13 ms
// of the first search of a process.  A context's creation starts ONE background thread per process and device that asks for
// the attributes of a kernel of every unit (which loads the unit) while the host parses its table; a launch that needs a
// unit before the thread got there loads it itself, as before.  CAFEHIP_PRELOAD=0 disables.
struct KernelPreload {
    std::mutex mu;
    std::thread th;
    bool started[64] = {};
    ~KernelPreload()
    {
        if (th.joinable()) th.join();
    }
    void start(int device)
    {
        const char* e = getenv("CAFEHIP_PRELOAD");
        if (e && atoi(e) == 0) return;
        std::lock_guard<std::mutex> g(mu);
        if (device < 0 || device >= 64 || started[device]) return;
        started[device] = true;
        if (th.joinable()) th.join();
        th = std::thread([device] {
            if (hipSetDevice(device) != hipSuccess) return;
            const void* fns[] = {k1_rb_kernel(), k2c_gemm_kernel(1, 1, 1, 512), k2_mfma4_objective_kernel(1, 1), k2_mfma4_small_r_kernel(1, 1), k3_kernel(true), k2_mfma16_objective_kernel(1, 1),
                                 k2_mfma4_kernel(1, 1), k2_mfma16_kernel(1, 1),
                                 k2c_kernel(1, 1, true, false)};
            for (const void* fn : fns) {
                hipFuncAttributes at;
                if (fn) (void)hipFuncGetAttributes(&at, fn);
            }
            (void)hipGetLastError();
        });
    }
    void join()
    {
        std::lock_guard<std::mutex> g(mu);
        if (th.joinable()) th.join();
    }
};
KernelPreload& kernel_preload()
{
    static KernelPreload p;
    return p;
}
}  // namespace

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
    kernel_preload().start(device_id);
    *out = c;
    return 0;
}

void cafehip_destroy(cafehip_ctx* c)
{
    if (!c) return;
    kernel_preload().join();
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
        if (c->opt.test_stall_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(c->opt.test_stall_ms));   // (tests: a host kept off its core)
        const int32_t want = c->eval_seq;
        // A chain armed behind this evaluation repeats it (its block is a copy of this one's) and publishes the NEXT number:
        // if this thread was kept off its core for longer than the gate waits (20 ms: a loaded host), the repetition has run
        // and its number stands where ours was -- same chunk sums, same first-zero index, same per-family values.  Accept it
        // (found by the round-5 soak run: "score kernel finished without publishing its result").
        bool repetition_ran = false;
        auto published = [&]() {
            const int32_t d = c->h_result->done_seq;
            if (d == want) return true;
            if (c->armed.on && d == c->armed.seq) {
                repetition_ran = true;
                return true;
            }
            return false;
        };
        unsigned long spins = 0;
        while (!published()) {
            if ((++spins & 0x3FFFF) == 0) {
                hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) {
                    if (!published()) HIP_TRY(hipStreamSynchronize(c->stream));
                    if (!published()) return fail("score kernel finished without publishing its result");
                    break;
                }
                if (q != hipErrorNotReady) return fail("stream error while waiting: %s", hipGetErrorString(q));
            }
        }
        if (repetition_ran) {
            c->armed.on = false;   // (it is spent)
            ++c->prearm_wasted;
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
        snprintf(buf, sizeof buf, " compressed(nodes=%d levels=%zu states=%ld top_states=%ld walk_steps=%zu walk_cols=%d used=%d level_tiles=", c->cp.n_nodes,
                 c->cp.level_first.size() - 1, c->cp.states, c->cp.top_states, c->cp.sched.ops.size(), c->cp.n_cols, (int)c->last_compressed);
        c->desc += buf;
        for (size_t l = 0; l + 1 < c->cp.level_first.size(); ++l) c->desc += (l ? "/" : "") + std::to_string(c->cp.level_first[l + 1] - c->cp.level_first[l]);
        c->desc += ")";
    }
    return c->desc.c_str();
}

}  // extern "C"
