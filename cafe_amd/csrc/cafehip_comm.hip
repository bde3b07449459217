// cafehip_comm.hip -- the multi-GPU entry points of the C ABI (include/cafehip.h, cafehip_comm_*): communicator set-up with
// its functional probe and mode agreement, block wiring / re-alignment, the sharded objective evaluation with its
// host-paced re-poll, host all-gather, status.  The communicator itself is comm.hpp; the evaluation it wraps is
// eval_device (cafehip.hip).  No reference counterpart: the reference is one process (cafe/lambda.cpp:698-722 is the
// map + sum being sharded).
#include "context.hpp"

extern "C" {

// ---- multi-GPU exchange behind the ABI (comm.hpp) ---------------------------------------------------------------
int cafehip_comm_unique_id(void* out_id)
{
    if (!out_id) return fail("null argument");
    FILE* f = fopen("/dev/urandom", "rb");
    const size_t got = f ? fread(out_id, 1, CAFEHIP_COMM_ID_BYTES, f) : 0;
    if (f) fclose(f);
    if (got != CAFEHIP_COMM_ID_BYTES) {
        // no entropy source: time and pid are unique enough for a rendezvous name on one node
        unsigned long long v[CAFEHIP_COMM_ID_BYTES / 8];
        const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
        for (size_t i = 0; i < CAFEHIP_COMM_ID_BYTES / 8; ++i) v[i] = t * (2 * i + 1) ^ ((unsigned long long)getpid() << 17) ^ (i * 0x9E3779B97F4A7C15ull);
        memcpy(out_id, v, CAFEHIP_COMM_ID_BYTES);
    }
    return 0;
}

// Functional probe of the peer mappings (k_x_probe): this rank's kernel stores into every peer's probe words and waits
// <= 1 s for theirs.  Returns how many peers' stores arrived here (world: all).
static int run_comm_probe(cafehip_ctx* c, CommLink& L)
{
    L.probe_ran = false;
    L.peers_seen = 0;
    if (!L.p2p_ok) return 0;   // some rank could not even map: nobody launches (the peers' words would never be written)
    const char* inj = getenv("CAFEHIP_COMM_INJECT");   // tests: "mute:<rank>" -- mapped, but its stores never leave
    int mute = 0;
    if (inj && !strncmp(inj, "mute:", 5) && atoi(inj + 5) == L.rank) mute = 1;
    c->comm_injected = mute;
    int32_t* d_seen = nullptr;
    HIP_TRY(hipMalloc(&d_seen, sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(d_seen, 0, sizeof(int32_t), c->stream));
    XProbeArgs a;
    memset(&a, 0, sizeof a);
    for (int r = 0; r < L.world; ++r) a.probe[r] = reinterpret_cast<unsigned long long*>(CommLink::probe_of(L.peer_xbuf[r]));
    a.rank = L.rank;
    a.world = L.world;
    a.mute = mute;
    a.nonce = L.nonce;
    a.timeout_ticks = (long long)(std::min(1.0, comm_timeout_s()) * 1e8);
    a.seen = d_seen;
    // Every rank's wait slice (<= 1 s) counts from ITS kernel's start, so the kernels must start together: the kernel's code
    // object is loaded by a launch that does nothing (world 0: no stores, no waits), the stream drained, and the ranks meet
    // at a host barrier immediately before the real launch -- a rank that was still allocating or loading code would
    // otherwise make a healthy fabric fail the early rank's probe.
    {
        XProbeArgs warm = a;
        warm.world = 0;
        if (launch_kernel(kx_probe_kernel(), dim3(1), dim3(64), 0, c->stream, warm) || hipStreamSynchronize(c->stream) != hipSuccess) {
            hipFree(d_seen);
            return fail("communicator: the probe kernel could not be launched");
        }
        HIP_TRY(hipMemsetAsync(d_seen, 0, sizeof(int32_t), c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (!L.barrier()) {
            hipFree(d_seen);
            return fail("communicator: %s", L.error.c_str());
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (launch_kernel(kx_probe_kernel(), dim3(1), dim3(64), 0, c->stream, a)) {
        hipFree(d_seen);
        return -1;
    }
    int32_t seen = 0;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(&seen, d_seen, sizeof seen, hipMemcpyDeviceToHost);
    hipFree(d_seen);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        seen = 0;   // a faulting probe is a failed probe: the ranks fall back together
    }
    L.probe_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    L.probe_ran = true;
    L.peers_seen = seen;
    return seen;
}

int cafehip_comm_init(cafehip_ctx* c, int rank, int world, const void* unique_id)
{
    if (!c || !unique_id) return fail("null argument");
    if (c->link) return fail("this context already belongs to a communicator");
    HIP_TRY(hipSetDevice(c->device));
    CommLink* L = new CommLink();
    if (!L->init(c->device, rank, world, unique_id)) {
        const std::string msg = L->error;
        delete L;
        return fail("communicator: %s", msg.c_str());
    }
    // "mapped" is not "reachable": every rank probes its peers with real stores and loads, and the ranks agree on ONE
    // mode -- direct only if every rank saw every peer, else RCCL, else the communicator fails on every rank -- here,
    // not inside the first evaluation.  (Everybody has zeroed its buffer and passed two barriers since: setup_p2p.)
    const int seen = run_comm_probe(c, *L);
    if (seen < 0) {
        L->fail(cafehip_last_error());
        delete L;
        return -1;
    }
    const int mode = L->decide_mode(seen == world, c->comm_mode != 2, [&] { return L->ensure_rccl(); });
    if (mode <= 0) {
        const std::string msg = mode < 0 ? L->error
                                         : "no exchange mode works on every rank: direct refused (this rank mapped " + std::to_string(L->peers_mapped) +
                                               " and heard " + std::to_string(L->peers_seen) + " of " + std::to_string(world) +
                                               " ranks; at least one rank did not hear all)" +
                                               (c->comm_mode == 2 ? ", and option comm=direct rules out RCCL" : ", RCCL: " + (L->error.empty() ? std::string("unavailable on some rank") : L->error));
        delete L;
        return fail("communicator: %s", msg.c_str());
    }
    if (mode == 1 && world > 1)
        fprintf(stderr, "cafehip: rank %d: direct exchange refused by the probe (mapped %d, heard %d of %d ranks) -- all ranks use RCCL\n", rank,
                L->peers_mapped, L->peers_seen, world);
    c->comm_agreed_mode = mode;
    c->link = L;
    c->x_seq = 0;
    c->blk_lo.clear();
    c->blk_hi.clear();
    return 0;
}

// exchange mode a sharded evaluation will use: 2 direct (peer buffers mapped on every rank) unless RCCL was asked for
static int comm_pick_mode(cafehip_ctx* c)
{
    if (c->comm_mode == 1) return 1;
    if (c->link->direct_ok) return 2;   // the verdict every rank agreed on after the functional probe
    return c->comm_mode == 2 ? -1 : 1;
}

// collective: everybody is between evaluations; clear my exchange buffer between two barriers and restart the sequence
static int comm_realign(cafehip_ctx* c)
{
    CommLink& L = *c->link;
    disarm(c);   // (a pre-armed chain of the single-GPU path would hold the stream for its slice)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!L.barrier()) return fail("communicator: %s", L.error.c_str());
    if (L.xbuf) {
        HIP_TRY(hipMemset(L.xbuf, 0, 2 * CommLink::parity_stride_bytes()));
        HIP_TRY(hipDeviceSynchronize());
    }
    c->x_seq = 0;   // ranks that fell out of step (one failed or skipped an evaluation) are in step again from here
    if (!L.barrier()) return fail("communicator: %s", L.error.c_str());
    return 0;
}

int cafehip_comm_set_blocks(cafehip_ctx* c, const int32_t* block_lo, const int32_t* block_hi)
{
    if (!c || !block_lo || !block_hi) return fail("null argument");
    if (!c->link) return fail("cafehip_comm_init has not been called");
    CommLink& L = *c->link;
    HIP_TRY(hipSetDevice(c->device));
    int slots = 1;
    for (int r = 0; r < L.world; ++r) {
        if (block_hi[r] < block_lo[r] || (r > 0 && block_lo[r] != block_hi[r - 1]) ||
            (block_lo[r] % CAFEHIP_CHUNK != 0 && block_hi[r] != block_lo[r]))   // (an empty block may sit at the table's ragged end)
            return fail("rank %d: block [%d, %d) must be contiguous with its neighbour's and start on a multiple of %d", r, block_lo[r], block_hi[r], CAFEHIP_CHUNK);
        slots = std::max(slots, (block_hi[r] - block_lo[r] + CAFEHIP_CHUNK - 1) / CAFEHIP_CHUNK);
    }
    if (block_hi[L.rank] - block_lo[L.rank] != c->F)
        return fail("this rank's block holds %d families but its table has %d", block_hi[L.rank] - block_lo[L.rank], c->F);
    if (slots > kCommSlotCap) return fail("%d chunks per rank exceed the exchange buffer (%d)", slots, kCommSlotCap);
    c->blk_lo.assign(block_lo, block_lo + L.world);
    c->blk_hi.assign(block_hi, block_hi + L.world);
    c->x_slots = slots;
    HIP_TRY(hipStreamSynchronize(c->stream));
    // host mirror: world rows of slots + 1 doubles
    const size_t need = (size_t)L.world * (slots + 1);
    if (!c->h_result || need > c->h_result_chunks) {
        hipHostFree(c->h_result);
        c->h_result = nullptr;
        const size_t bytes = sizeof(HostResult) + need * sizeof(double);
        HIP_TRY(hipHostMalloc((void**)&c->h_result, bytes, hipHostMallocMapped | hipHostMallocCoherent));
        memset((void*)c->h_result, 0, bytes);
        c->h_result_chunks = need;
        c->host_seq = 0;
        c->eval_seq = 0;
    }
    // direct mode: rows of ranks with fewer chunks must read 0 in the slots they never write.  Everybody is between
    // evaluations here (collective call): clear my buffer between two barriers, sequence numbers back to 0
    if (comm_realign(c)) return -1;
    // RCCL mode buffers
    if (slots != c->packed_slots || !c->d_packed) {
        hipFree(c->d_packed);
        hipFree(c->d_gathered);
        c->d_packed = c->d_gathered = nullptr;
        HIP_TRY(hipMalloc(&c->d_packed, (size_t)(slots + 1) * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_gathered, need * sizeof(double)));
        c->packed_slots = slots;
    }
    HIP_TRY(hipMemset(c->d_packed, 0, (size_t)(slots + 1) * sizeof(double)));   // unused chunk slots read 0 on every rank
    HIP_TRY(hipDeviceSynchronize());
    if (!L.barrier()) return fail("communicator: %s", L.error.c_str());
    return 0;
}

static int wait_host_seq(cafehip_ctx* c, int32_t want, bool* peer_timeout)
{
    // spin on the sequence number the last score block publishes (a few microseconds after the kernel ends); fall
    // back to a stream query now and then so that a faulted launch cannot hang us
    unsigned long spins = 0;
    if (peer_timeout) *peer_timeout = false;
    for (;;) {
        const int32_t seen = c->h_result->done_seq;
        if (seen == want) break;
        if (peer_timeout && seen == -want) {
            *peer_timeout = true;
            break;
        }
        if ((++spins & 0x3FFFF) == 0) {
            hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (c->h_result->done_seq != want && !(peer_timeout && c->h_result->done_seq == -want)) HIP_TRY(hipStreamSynchronize(c->stream));
                const int32_t now = c->h_result->done_seq;
                if (peer_timeout && now == -want) *peer_timeout = true;
                else if (now != want) return fail("score kernel finished without publishing its result");
                break;
            }
            if (q != hipErrorNotReady) return fail("stream error while waiting: %s", hipGetErrorString(q));
        }
    }
    // the payload was written before the sequence number (device-side system fence): order our reads after the flag read
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}

int cafehip_eval_posterior_sharded(cafehip_ctx* c, const double* node_lambda, const double* node_mu, const double* prior,
                                   double* score, int32_t* first_zero_global)
{
    if (!c) return fail("null context");
    if (!node_lambda || !node_mu || !prior || !score) return fail("null argument");
    if (!c->link || c->blk_lo.empty()) return fail("cafehip_comm_init / cafehip_comm_set_blocks have not been called");
    if (c->d_err && c->err_mfs < c->range_max)
        return fail("error model covers sizes 0..%d but range_max is %d", c->err_mfs, c->range_max);
    CommLink& L = *c->link;
    if (c->blk_hi[L.rank] - c->blk_lo[L.rank] != c->F) return fail("the table changed: call cafehip_comm_set_blocks again");
    const int mode = comm_pick_mode(c);
    if (mode < 0) return fail("direct exchange asked for but the peer buffers could not be mapped on every rank");
    const int slots = c->x_slots, row_len = slots + 1;
    const double* rows = nullptr;
    if (mode == 2) {
        if (eval_device(c, node_lambda, node_mu, prior, nullptr, c->d_first_zero, true, 1, true)) return -1;
        bool peer_timeout = false;
        if (wait_host_seq(c, c->host_seq, &peer_timeout)) return -1;
        const auto t_wait0 = std::chrono::steady_clock::now();
        while (peer_timeout) {
            // the score kernel gave up after one slice: wait on in slices of the same length (k_x_collect, one workgroup)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() + x_wait_slice_s() > comm_timeout_s())
                return fail("direct exchange: a rank did not deliver its row within %.0f s", comm_timeout_s());
            K3xArgs x = c->x_last;
            x.seq = ++c->host_seq;
            ++c->x_repolls;
            if (launch_kernel(kx_collect_kernel(), dim3(1), dim3(CAFEHIP_CHUNK), 0, c->stream, x)) return -1;
            if (wait_host_seq(c, c->host_seq, &peer_timeout)) return -1;
        }
        rows = c->h_result->chunk_sums;
    } else {
        if (!L.rccl && !L.ensure_rccl()) return fail("RCCL exchange: %s", L.error.c_str());
        int32_t* d_fz = reinterpret_cast<int32_t*>(c->d_packed + slots);
        if (eval_device(c, node_lambda, node_mu, prior, c->d_packed, d_fz)) return -1;
        const auto t0 = std::chrono::steady_clock::now();
        if (c->timing) {
            if (!c->ev_x0) {
                HIP_TRY(hipEventCreate(&c->ev_x0));
                HIP_TRY(hipEventCreate(&c->ev_x1));
            }
            HIP_TRY(hipEventRecord(c->ev_x0, c->stream));
        }
        // the one exchange step: ONE ncclAllGather of the packed rows on the context's stream, picked up without a
        // copy command or a stream synchronisation
        const int rc = rccl_api().AllGather(c->d_packed, c->d_gathered, (size_t)row_len, ncclDouble, L.rccl, c->stream);
        if (rc != ncclSuccess) return fail("ncclAllGather: %s", rccl_api().GetErrorString(rc));
        const void* host = nullptr;
        if (cafehip_fetch_small(c, c->d_gathered, (size_t)row_len * L.world * sizeof(double), &host)) return -1;
        if (c->timing) {
            HIP_TRY(hipEventRecord(c->ev_x1, c->stream));
            HIP_TRY(hipEventSynchronize(c->ev_x1));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev_x0, c->ev_x1));
            c->last_exchange_ms = ms;
        }
        c->x_host_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rows = static_cast<const double*>(host);
    }
    ++c->x_calls;
    c->x_mode_used = mode;
    if (collect_kernel_ms(c)) return -1;
    // the same fixed-order sum on every rank: chunk order == family order; empty slots add 0
    double s = 0.0;
    int fz = -1;
    for (int r = 0; r < L.world; ++r) {
        const double* row = rows + (size_t)r * row_len;
        for (int k = 0; k < slots; ++k) s += row[k];
        long long local;
        if (mode == 2) {
            memcpy(&local, row + slots, sizeof local);
        } else {
            int32_t l32;
            memcpy(&l32, row + slots, sizeof l32);
            local = l32;
        }
        if (local >= 0 && local < c->blk_hi[r] - c->blk_lo[r] && fz < 0) fz = c->blk_lo[r] + (int)local;
    }
    *score = fz >= 0 ? -INFINITY : s;   // cafe/lambda.cpp:753-760
    if (first_zero_global) *first_zero_global = fz;
    return 0;
}

int cafehip_comm_resync(cafehip_ctx* c)
{
    if (!c) return fail("null context");
    if (!c->link) return fail("cafehip_comm_init has not been called");
    HIP_TRY(hipSetDevice(c->device));
    return comm_realign(c);
}

int cafehip_comm_status(cafehip_ctx* c, int32_t out[CAFEHIP_COMM_STATUS_WORDS], double* probe_ms)
{
    if (!c || !out) return fail("null argument");
    memset(out, 0, sizeof(int32_t) * CAFEHIP_COMM_STATUS_WORDS);
    if (probe_ms) *probe_ms = 0.0;
    if (!c->link) return 0;
    const CommLink& L = *c->link;
    out[0] = L.world;
    out[1] = c->comm_agreed_mode;
    out[2] = c->x_mode_used ? c->x_mode_used : std::max(comm_pick_mode(c), 0);
    out[3] = L.direct_ok ? 1 : 0;
    out[4] = L.peers_mapped;
    out[5] = L.probe_ran ? L.peers_seen : -1;
    out[6] = L.rccl != nullptr ? 1 : 0;
    out[7] = L.rccl_count;
    out[8] = c->comm_injected;
    out[9] = (int32_t)std::min<long>(c->x_repolls, INT32_MAX);
    if (probe_ms) *probe_ms = L.probe_ms;
    return 0;
}

int cafehip_comm_cleanup(const void* unique_id)
{
    if (!unique_id) return fail("null argument");
    return CommLink::unlink_names(unique_id);
}

int cafehip_comm_mode_selftest(int rank, int world, const void* unique_id, int my_probe_ok, int my_rccl_ok, int* mode)
{
    // the mode agreement alone (CommLink::decide_mode over the shared-memory mailboxes), the local outcomes injected:
    // what the CPU test suite runs with several processes, one of them "mapped but unreachable"
    if (!unique_id || !mode) return fail("null argument");
    CommLink L;
    if (!L.init(-1, rank, world, unique_id)) return fail("communicator: %s", L.error.c_str());
    // (my_rccl_ok: 1 joins, 0 fails to join, -1 ruled out by a local option: then NO rank may enter the collective join)
    *mode = L.decide_mode(my_probe_ok != 0, my_rccl_ok >= 0, [&] { return my_rccl_ok > 0; });
    if (*mode < 0) return fail("communicator: %s", L.error.c_str());
    return L.barrier() ? 0 : fail("communicator: %s", L.error.c_str());
}

int cafehip_comm_allgather(cafehip_ctx* c, const void* mine, size_t nbytes_mine, void* all, size_t nbytes_slot)
{
    if (!c || !all) return fail("null argument");
    if (!c->link) return fail("cafehip_comm_init has not been called");
    if (nbytes_mine > nbytes_slot) return fail("block of %zu bytes does not fit its %zu-byte slot", nbytes_mine, nbytes_slot);
    if (!c->link->host_allgather(mine, nbytes_mine, all, nbytes_slot)) return fail("communicator: %s", c->link->error.c_str());
    return 0;
}

int cafehip_comm_info(cafehip_ctx* c, int* rank, int* world, int* mode, double* exchange_ms, double* host_seconds, long* calls)
{
    if (!c) return fail("null context");
    if (rank) *rank = c->link ? c->link->rank : 0;
    if (world) *world = c->link ? c->link->world : 1;
    if (mode) *mode = c->link ? (c->x_mode_used ? c->x_mode_used : std::max(comm_pick_mode(c), 0)) : 0;
    if (exchange_ms) *exchange_ms = c->last_exchange_ms;
    if (host_seconds) *host_seconds = c->x_host_seconds;
    if (calls) *calls = c->x_calls;
    return 0;
}

int cafehip_comm_host_selftest(int rank, int world, const void* unique_id, const void* mine, size_t nbytes_mine, void* all,
                               size_t nbytes_slot)
{
    // the host half of the communicator alone (rendezvous, mailboxes, barrier, host all-gather): no context, no
    // device needed -- exercised by the CPU test suite with several processes
    if (!unique_id || !all) return fail("null argument");
    CommLink L;
    if (!L.init(-1, rank, world, unique_id)) return fail("communicator: %s", L.error.c_str());
    for (int round = 0; round < 3; ++round)
        if (!L.barrier()) return fail("communicator: %s", L.error.c_str());
    if (!L.host_allgather(mine, nbytes_mine, all, nbytes_slot)) return fail("communicator: %s", L.error.c_str());
    return L.barrier() ? 0 : fail("communicator: %s", L.error.c_str());
}

}  // extern "C"
