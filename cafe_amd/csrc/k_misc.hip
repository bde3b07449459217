// k_misc.hip -- the kernels that are not matrix-core walks:
//   k2_prune_v1   row-per-thread pruning (vector FMA, k ascending as libtree/birthdeath.c:173-180): fallback for matrix
//                 sides beyond the matrix-core wave grids and A/B runs
//   k3_score, k3_cluster_score   per-chunk sums of log max-posterior in family order + first zero-likelihood family
//                 == get_posterior, cafe/lambda.cpp:691-724; cafe_get_clustered_posterior, cafe/cafe_main.c:165-253
//   k_fetch_small device words -> pinned host mirror + sequence number
//   k4_viterbi    max-product walk + backtrack == cafe_tree_viterbi, cafe/viterbi.cpp:208-351
#include <climits>
#include <cmath>

#include "kernels.hpp"
#include "k3_device.hpp"

namespace {
using namespace cafehip;

// ------------------------------------------------------------------------------------
// K2 (v1, vector FMA): one workgroup carries NF families through the whole tree.
// Thread r owns output row r of every node vector; node vectors live in LDS slots
// [slot][fam][LDv].  Per child edge the thread streams its column of the transposed
// matrix PT[k][row_lo + r] (coalesced across the workgroup, L2 resident, shared by
// all families) and accumulates NF dot products with the child's vectors, which are
// LDS broadcasts.  k runs ascending, i.e. in the reference's summation order
// (libtree/birthdeath.c:173-180).  A one-hot leaf (cafe/cafe_tree.c:208-209) turns
// the product into the gather PT[count][row].
// REF (option k2=v1ref): the REFERENCE's arithmetic, not only its order -- every term a separate multiplication and a separate
// addition (square_matrix_multiply, libtree/birthdeath.c:163-182, as gcc builds it for x86-64: no fused multiply-add), the
// vector of a node the product of its two factors (cafe/cafe_tree.c:261-266).  On matrices built in K1's exact form (the host
// libm's exp(), exp_like_host.hpp) the node vectors then carry the reference build's bits: what the report phase can ask for
// when its comparisons -- a likelihood's rank in a sorted null -- are to be the reference's for every input, at the price of
// the vector unit's speed.
// ------------------------------------------------------------------------------------
template <int NF, bool REF>
__global__ __launch_bounds__(1024) void k2_prune_v1(K2Args a)
{
    // dynamic LDS: [n_slots (+1 with an error model)][NF][LDv] node vectors, then the tile's counts and column limits
    extern __shared__ double smem[];
    int* const s_cnt = reinterpret_cast<int*>(smem + (size_t)(a.n_slots + (a.err ? 1 : 0)) * NF * a.LDv);   // [NF][n_leaves]
    int* const s_colmax = s_cnt + NF * a.n_leaves;                                                            // [NF]

    const int tid = threadIdx.x;
    const int r = tid;
    const int fam0 = blockIdx.x * NF;
    const size_t slot_stride = (size_t)NF * a.LDv;
    const bool batch = (a.col_max != nullptr);

    for (int i = tid; i < NF * a.n_leaves; i += blockDim.x) {
        const int f = i / a.n_leaves, j = i - f * a.n_leaves;
        const int u = fam0 + f;
        s_cnt[f * a.n_leaves + j] = (u < a.Fu) ? a.counts[(size_t)u * a.n_leaves + j] : 0;
    }
    if (tid < NF) {
        const int u = fam0 + tid;
        s_colmax[tid] = (batch && u < a.Fu) ? a.col_max[u] : (a.C - 1);
    }
    __syncthreads();

    const int err_slot = a.n_slots;  // scratch slot for error-model leaf vectors
    int root_slot = 0;

    for (int oi = 0; oi < a.n_ops; ++oi) {
        const cafehip::PruneOp op = a.ops[oi];
        const int rows = op.is_root ? a.R : a.C;
        const int row_lo = op.is_root ? a.root_min : 0;
        double y[2][NF];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const double* PTc =
                a.PT + (size_t)a.node_key[op.child[ch]] * a.KP * a.LD + row_lo + r;
            const bool errleaf =
                (op.kind[ch] == 0) && a.err != nullptr && a.leaf_has_err[op.src[ch]];
            if (op.kind[ch] == 0 && !errleaf) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int cnt = s_cnt[f * a.n_leaves + op.src[ch]];
                    y[ch][f] = (r < rows && cnt <= s_colmax[f]) ? PTc[(size_t)cnt * a.LD] : 0.0;
                }
            } else {
                const double* src;
                if (errleaf) {
                    // leaf vector = errormatrix[observed][0..C) (cafe/cafe_tree.c:196-203)
                    double* es = smem + (size_t)err_slot * slot_stride;
                    __syncthreads();
                    for (int i = tid; i < NF * a.LDv; i += blockDim.x) {
                        const int f = i / a.LDv, k = i - f * a.LDv;
                        const int cnt = s_cnt[f * a.n_leaves + op.src[ch]];
                        es[i] = (k < a.C && k <= s_colmax[f]) ? a.err[(size_t)cnt * a.err_ld + k] : 0.0;
                    }
                    __syncthreads();
                    src = es;
                } else {
                    src = smem + (size_t)op.src[ch] * slot_stride;
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) y[ch][f] = 0.0;
                if (r < rows) {
                    if constexpr (REF) {
#pragma clang fp contract(off)
                        for (int k = 0; k < a.C; ++k) {   // (k < C exactly: the padding column is not a term of the reference's sum)
                            const double p0 = PTc[(size_t)k * a.LD];
#pragma unroll
                            for (int f = 0; f < NF; ++f) {
                                const double term = p0 * src[f * a.LDv + k];
                                y[ch][f] = y[ch][f] + term;
                            }
                        }
                    } else {
                        for (int k = 0; k < a.C; k += 2) {
                            const double p0 = PTc[(size_t)k * a.LD];
                            const double p1 = PTc[(size_t)(k + 1) * a.LD];
#pragma unroll
                            for (int f = 0; f < NF; ++f) {
                                const double2 l = *reinterpret_cast<const double2*>(src + f * a.LDv + k);
                                y[ch][f] = fma(p0, l.x, y[ch][f]);
                                y[ch][f] = fma(p1, l.y, y[ch][f]);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();  // every read of the source slots is done: dst may alias a source
        double* dst = smem + (size_t)op.dst * slot_stride;
        for (int rr = tid; rr < a.LDv; rr += blockDim.x) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                double v = 0.0;
                if (rr == r && r < rows) {
                    v = y[0][f] * y[1][f];
                    // rows beyond this family's column range do not exist in the reference
                    // (range.max is per call there); zero them so they add exact zeros upstream
                    if (!op.is_root && r > s_colmax[f]) v = 0.0;
                }
                dst[f * a.LDv + rr] = v;
            }
        }
        __syncthreads();
        root_slot = op.dst;
    }

    // ---- root vector -> posterior (cafe/lambda.cpp:657-689) or packed root rows ----
    const double* Lr = smem + (size_t)root_slot * slot_stride;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    for (int f = wave; f < NF; f += nwaves) {
        const int u = fam0 + f;
        if (u >= a.Fu) continue;
        const double* L = Lr + f * a.LDv;
        if (batch) {
            const int lo = a.root_lo[u] - a.root_min, hi = a.root_hi[u] - a.root_min;
            double* o = a.out_root + a.out_off[u];
            for (int i = lo + lane; i <= hi; i += 64) o[i - lo] = L[i];
            continue;
        }
        double best = -INFINITY, bestp = -INFINITY;
        int bi = INT_MAX;  // INT_MAX = this lane has seen no element yet
        for (int i = lane; i < a.R; i += 64) {
            const double v = L[i];
            if (bi == INT_MAX || v > best) {
                best = v;
                bi = i;
            }
            const double p = exp(log(v) + a.logprior[i]);
            bestp = fmax(bestp, p);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off);
            const int oi2 = __shfl_xor(bi, off);
            const double op2 = __shfl_xor(bestp, off);
            // first maximum wins (libcommon/mathfunc.c:9-24): larger value, then lower index
            if (oi2 != INT_MAX && (bi == INT_MAX || ov > best || (ov == best && oi2 < bi))) {
                best = ov;
                bi = oi2;
            }
            bestp = fmax(bestp, op2);
        }
        if (lane == 0) {
            a.max_lik[u] = best;
            a.argmax[u] = bi;
            a.max_post[u] = bestp;
        }
    }
}

// ------------------------------------------------------------------------------------
// K3: score.  One workgroup per chunk of CAFEHIP_CHUNK families in FAMILY order
// (duplicates expanded through fam2u), fixed-shape tree sum -> chunk_sums[chunk].
// ------------------------------------------------------------------------------------
// Results of the synchronous path go straight to pinned, device-visible host memory (no copy kernels,
// no interrupt-driven wait): every block stores its chunk sum there, the last block to arrive (device
// counter) publishes the first-zero index and a sequence number the host spins on.
template <bool HOST_OUT>
__global__ __launch_bounds__(CAFEHIP_CHUNK) void k3_score(K3Args a)
{
    k3_score_block<HOST_OUT>(a, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}


// ------------------------------------------------------------------------------------
// K3 of a sharded evaluation, direct exchange (comm.hpp).  The packed row of a rank -- its chunk sums in family order
// and the index of its first zero-likelihood family -- is what cafe/lambda.cpp:698-722 needs from that rank's
// families.  Every block stores its chunk sum straight into the exchange buffer of EVERY rank (uncached device
// memory, the peers' mapped over xGMI with hipIpc: world 8-byte stores, each followed by a system-scope fence before
// the block counts itself in); the last block publishes the first-zero index the same way, then raises this rank's
// flag (= the exchange sequence number) in every buffer, waits until every rank's flag stands in its own buffer and
// copies all rows to the pinned host block the host spins on.  No collective launch, no extra kernel: the sharded
// evaluation is the same three launches as the single-GPU one.  Buffers alternate by the parity of the sequence
// number: a rank can be at most one evaluation ahead of another (it needs the other's row to finish).
// The wait is bounded (timeout_ticks, ~1 s): a GPU never sits in this kernel longer; the host reads -seq and re-polls
// with k_x_collect in slices until ITS patience (comm_timeout_s) is over.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long x_load_flag(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(CAFEHIP_CHUNK) void k3_score_x(K3xArgs a)
{
    __shared__ double red[CAFEHIP_CHUNK];
    __shared__ int s_last, s_fail, s_fz;
    const int i = blockIdx.x * CAFEHIP_CHUNK + threadIdx.x;
    double v = 0.0;
    if (i < a.F) {
        const int u = a.fam2u ? a.fam2u[i] : i;
        v = log(a.max_post_u[u]);                                     // cafe/lambda.cpp:721
        if (a.max_lik_u[u] == 0.0) atomicMin(a.first_zero, i);        // cafe/lambda.cpp:715-720
    }
    red[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = CAFEHIP_CHUNK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const size_t row = (size_t)a.rank * (a.slots + 1);
    if ((int)threadIdx.x < a.world) {
        a.rows[threadIdx.x][row + blockIdx.x] = red[0];
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(a.arrive, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) {
        s_fail = 0;
        s_fz = atomicExch(a.first_zero, INT32_MAX);   // atomic read of the final value; the word is left reset for the next evaluation
    }
    __syncthreads();
    const long long fz = s_fz;
    if ((int)threadIdx.x < a.world) {
        a.rows[threadIdx.x][row + a.slots] = __longlong_as_double(fz);
        __threadfence_system();
        __hip_atomic_store(&a.flags[threadIdx.x][a.rank], a.xseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) *a.arrive = 0;
    __syncthreads();
    if ((int)threadIdx.x < a.world) {
        const unsigned long long* mine = a.flags[a.rank] + threadIdx.x;
        const long long t0 = wall_clock64();
        while (x_load_flag(mine) != a.xseq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > a.timeout_ticks) {
                s_fail = 1;
                break;
            }
        }
    }
    __syncthreads();
    const double* all = a.rows[a.rank];
    const int n = a.world * (a.slots + 1);
    if (!s_fail)
        for (int k = threadIdx.x; k < n; k += CAFEHIP_CHUNK) a.host->chunk_sums[k] = __builtin_nontemporal_load(all + k);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) a.host->done_seq = s_fail ? -a.seq : a.seq;   // -seq: the host re-polls with k_x_collect
}

// ------------------------------------------------------------------------------------
// The wait of k3_score_x on its own (one workgroup): a rank whose peers had not delivered within the score kernel's
// bounded wait (~1 s: a GPU never sits in one kernel longer than that) is re-polled by the HOST with this kernel until
// the host's own patience (comm_timeout_s) runs out -- a peer that is merely late (it wrote a report, loaded a table)
// is waited for in slices, a dead one ends the call.  Publishes seq (all rows in host->chunk_sums) or -seq.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(CAFEHIP_CHUNK) void k_x_collect(K3xArgs a)
{
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    if ((int)threadIdx.x < a.world) {
        const unsigned long long* mine = a.flags[a.rank] + threadIdx.x;
        const long long t0 = wall_clock64();
        while (x_load_flag(mine) != a.xseq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > a.timeout_ticks) {
                s_fail = 1;
                break;
            }
        }
    }
    __syncthreads();
    const double* all = a.rows[a.rank];
    const int n = a.world * (a.slots + 1);
    if (!s_fail)
        for (int k = threadIdx.x; k < n; k += CAFEHIP_CHUNK) a.host->chunk_sums[k] = __builtin_nontemporal_load(all + k);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) a.host->done_seq = s_fail ? -a.seq : a.seq;
}

// ------------------------------------------------------------------------------------
// Gate of a pre-armed launch chain (round 5).  The launches of the NEXT evaluation are queued behind this one-wave kernel
// while the current evaluation still runs; the host then starts them with ONE store to pinned memory instead of a launch
// (tools/gate_probe.hip: 3.9 us from the store to a word written by the released kernel, against 6.4 us for a launch).
// The wait is bounded: a host that never comes back (the loop ended, another entry point was called without disarming)
// costs one slice, not a hang -- the chain then runs on the parameter block it was armed with (a copy of the previous
// evaluation's: a valid, harmless repetition) and the outcome word tells the host that it did.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gate(GateArgs a)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    unsigned long long how = 2;
    for (;;) {
        if (__hip_atomic_load(a.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == a.want) {
            how = 1;
            break;
        }
        if (wall_clock64() - t0 > a.timeout_ticks) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(a.outcome, (a.want << 2) | how, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------
// Functional probe of the direct exchange, run once per communicator by cafehip_comm_init: lane t stores (nonce | my
// rank) into word [my rank] of rank t's probe area THROUGH THE PEER MAPPING and waits (bounded) until (nonce | t)
// stands in word [t] of my own area -- i.e. until rank t's store has really arrived in my memory and is visible to
// the same kind of load the score kernel polls with.  seen = peers whose word arrived (myself included).  `mute`
// (tests: CAFEHIP_COMM_INJECT) skips the stores: the others then see this rank as mapped but unreachable.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_x_probe(XProbeArgs a)
{
    __shared__ int s_seen;
    if (threadIdx.x == 0) s_seen = 0;
    __syncthreads();
    if ((int)threadIdx.x < a.world) {
        const int t = threadIdx.x;
        if (!a.mute) {
            __hip_atomic_store(&a.probe[t][a.rank], a.nonce | (unsigned long long)a.rank, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
        }
        const unsigned long long want = a.nonce | (unsigned long long)t;
        const unsigned long long* mine = a.probe[a.rank] + t;
        const long long t0 = wall_clock64();
        bool ok = false;
        for (;;) {
            if (x_load_flag(mine) == want) {
                ok = true;
                break;
            }
            if (wall_clock64() - t0 > a.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (ok) atomicAdd(&s_seen, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) *a.seen = s_seen;
}

// ------------------------------------------------------------------------------------
// K3 of the k-cluster model (cafe_get_clustered_posterior, cafe/cafe_main.c:165-253).  K2 has left the per-family
// max posterior of every cluster (set) in max_post_u[k * Fu + u].  Per family, clusters ascending as the reference
// loops them: MAP_k = max_post_k * weight_k (:196), sum (:197), membership p_z[k] = MAP_k / sum (:204),
// MAP = sum_k p_z[k] * MAP_k (:210-213); the score adds log(MAP) (:241) and the new weights are the mean memberships
// (:243-245).  One workgroup per chunk of CAFEHIP_CHUNK families in family order, fixed-shape tree sums for the
// score and for each cluster's membership; MAP == 0 marks the family (:231-240).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(CAFEHIP_CHUNK) void k3_cluster_score(K3cArgs a)
{
    const double* __restrict__ max_post_u = a.max_post_u;
    const int32_t* __restrict__ fam2u = a.fam2u;
    const int F = a.F, Fu = a.Fu, K = a.K;
    const ClusterWeights& cw = a.cw;
    double* __restrict__ chunk_sums = a.chunk_sums;
    double* __restrict__ memb_sums = a.memb_sums;
    int32_t* __restrict__ first_zero = a.first_zero;
    double* __restrict__ map_out = a.map_out;
    double* __restrict__ pz_out = a.pz_out;
    __shared__ double red[CAFEHIP_CHUNK];
    const int i = blockIdx.x * CAFEHIP_CHUNK + threadIdx.x;
    double pz[kMaxSets];
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < kMaxSets; ++k) pz[k] = 0.0;
    if (i < F) {
        const int u = fam2u[i];
        double mapk[kMaxSets];
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < kMaxSets; ++k) {
            mapk[k] = 0.0;
            if (k < K) {
                mapk[k] = max_post_u[(size_t)k * Fu + u] * cw.w[k];
                sum += mapk[k];
            }
        }
        double expected = 0.0;
#pragma unroll
        for (int k = 0; k < kMaxSets; ++k) {
            if (k < K) {
                pz[k] = mapk[k] / sum;
                expected += pz[k] * mapk[k];
                if (pz_out) pz_out[(size_t)i * K + k] = pz[k];
            }
        }
        if (map_out) map_out[i] = expected;
        if (expected == 0.0) atomicMin(first_zero, i);
        v = log(expected);
    }
    // score, then one tree sum per cluster membership
    for (int q = -1; q < K; ++q) {
        double x = v;
#pragma unroll
        for (int k = 0; k < kMaxSets; ++k)
            if (q == k) x = pz[k];
        __syncthreads();
        red[threadIdx.x] = x;
        __syncthreads();
#pragma unroll
        for (int s = CAFEHIP_CHUNK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (q < 0) chunk_sums[blockIdx.x] = red[0];
            else memb_sums[(size_t)q * gridDim.x + blockIdx.x] = red[0];
        }
    }
}

// device words -> pinned host mirror, then a sequence number (cafehip_fetch_small)
__global__ __launch_bounds__(256) void k_fetch_small(FetchArgs a)
{
    for (size_t i = threadIdx.x; i < a.n_words; i += 256) a.host_dst[i] = a.src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *a.host_seq = a.seq;
}

// ------------------------------------------------------------------------------------
// K4: Viterbi (cafe/viterbi.cpp:208-351).  Same walk as K2 with (max, argmax) in place of the
// sum: thread r owns row r, factor[r] = max_k PT[k][row] * L_child[k] with strict '>' over
// ascending k (first maximum wins), node vector = product of the two factors; the argmax
// tables of the internal children stay in LDS (16-bit) and one thread per family backtracks
// root -> leaves in prefix order.  No sums: every value is a chain of single multiplications,
// so given the same matrices the sizes equal a host evaluation bit for bit.
// ------------------------------------------------------------------------------------

template <int NF>
__global__ __launch_bounds__(1024) void k4_viterbi(K4Args a)
{
    extern __shared__ double smem4[];
    double* slots = smem4;                                                  // [n_slots][NF][LDv]
    // argmax tables [n_tables][NF][LDv]: written once per (node, row), read ~n_nodes times per family by the
    // backtrack -- in global scratch they cost next to no traffic and leave LDS to the node vectors, i.e. several
    // workgroups per CU instead of one (the k loop is latency-bound at one wave per SIMD)
    int* const s_cnt = reinterpret_cast<int*>(slots + (size_t)a.n_slots * NF * a.LDv);   // [NF][n_leaves]
    unsigned short* vit = a.vit_global
                              ? a.vit_global + (size_t)blockIdx.x * a.n_tables * NF * a.LDv
                              : reinterpret_cast<unsigned short*>(s_cnt + ((NF * a.n_leaves + 1) & ~1));
    __shared__ int s_colmax[NF];

    const int tid = threadIdx.x;
    const int r = tid;
    const int fam0 = blockIdx.x * NF;
    const size_t slot_stride = (size_t)NF * a.LDv;

    for (int i = tid; i < NF * a.n_leaves; i += blockDim.x) {
        const int f = i / a.n_leaves, j = i - f * a.n_leaves;
        const int u = fam0 + f;
        s_cnt[f * a.n_leaves + j] = (u < a.B) ? a.counts[(size_t)u * a.n_leaves + j] : 0;
    }
    if (tid < NF) s_colmax[tid] = (fam0 + tid < a.B) ? a.col_max[fam0 + tid] : (a.C - 1);
    if (!a.vit_global)   // (every entry the backtrack reads is written by the walk; the LDS copy is cleared for tidiness)
        for (int i = tid; i < a.n_tables * NF * a.LDv; i += blockDim.x) vit[i] = 0;
    __syncthreads();

    int root_slot = 0;
    for (int oi = 0; oi < a.n_ops; ++oi) {
        const cafehip::PruneOp op = a.ops[oi];
        const int rows = op.is_root ? a.R : a.C;
        const int row_lo = op.is_root ? a.root_min : 0;
        double y[2][NF];
        int arg[2][NF];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const double* PTc = a.PT + (size_t)a.node_key[op.child[ch]] * a.KP * a.LD + row_lo + r;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                y[ch][f] = 0.0;
                arg[ch][f] = 0;
            }
            if (op.kind[ch] == 0) {
                // one-hot leaf: the only non-zero product is at k = count
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int cnt = s_cnt[f * a.n_leaves + op.src[ch]];
                    if (r < rows && cnt <= s_colmax[f]) y[ch][f] = PTc[(size_t)cnt * a.LD];
                }
            } else if (r < rows) {
                const double* src = slots + (size_t)op.src[ch] * slot_stride;
                for (int k = 0; k < a.C; ++k) {
                    const double pv = PTc[(size_t)k * a.LD];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const double tmp = pv * src[f * a.LDv + k];
                        if (tmp > y[ch][f]) {   // cafe/viterbi.cpp:296-300
                            y[ch][f] = tmp;
                            arg[ch][f] = k;
                        }
                    }
                }
            }
            if (op.kind[ch] == 1 && r < rows) {
                const int tb = a.vit_slot[op.child[ch]];
#pragma unroll
                for (int f = 0; f < NF; ++f) vit[((size_t)tb * NF + f) * a.LDv + r] = (unsigned short)arg[ch][f];
            }
        }
        __syncthreads();
        double* dst = slots + (size_t)op.dst * slot_stride;
        for (int rr = tid; rr < a.LDv; rr += blockDim.x) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                double v = 0.0;
                if (rr == r && r < rows) {
                    v = y[0][f] * y[1][f];
                    if (!op.is_root && r > s_colmax[f]) v = 0.0;
                }
                dst[f * a.LDv + rr] = v;
            }
        }
        __syncthreads();
        root_slot = op.dst;
    }

    __threadfence_block();
    __syncthreads();
    // backtrack (cafe/viterbi.cpp:322-351): one thread per family, prefix order
    if (tid < NF && fam0 + tid < a.B) {
        const int f = tid;
        const int u = fam0 + f;
        int32_t* out = a.node_sizes + (size_t)u * a.n_nodes;
        const double* L = slots + (size_t)root_slot * slot_stride + f * a.LDv;
        const int lo = a.root_lo[u], hi = a.root_hi[u];
        for (int j = 0; j < a.n_leaves; ++j) out[2 * j] = s_cnt[f * a.n_leaves + j];
        int best = 0;
        if (hi >= lo) {
            double bv = L[lo - a.root_min];
            for (int s = lo + 1; s <= hi; ++s) {
                const double v = L[s - a.root_min];
                if (bv < v) {   // __maxidx: first maximum
                    bv = v;
                    best = s - lo;
                }
            }
        }
        out[a.root] = lo + best;
        for (int pi = 0; pi < a.n_nodes; ++pi) {
            const int node = a.prefix[pi];
            if (node == a.root || (node & 1) == 0) continue;   // leaves keep their counts
            const int par = a.parent[node];
            const int ps = out[par];
            int idx = ps;                      // base = range.min = 0
            int size;
            if (par == a.root) {
                // rows of a root child are indexed by root size; an empty root range computes none of
                // them in the reference (stale zeros)
                idx = ps - a.root_min;
                size = (hi >= lo && idx >= 0 && idx < a.R) ? vit[((size_t)a.vit_slot[node] * NF + f) * a.LDv + idx] : 0;
            } else {
                size = (idx >= 0 && idx < a.C) ? vit[((size_t)a.vit_slot[node] * NF + f) * a.LDv + idx] : 0;
            }
            out[node] = size;
        }
    }
}

}  // namespace

namespace cafehip {

const void* k2_v1_kernel(int nf, bool reference_arithmetic)
{
    if (reference_arithmetic) {
        switch (nf) {
            // (16 families per workgroup needs scratch in this form: not instantiated, the launcher stops at 8)
            case 8: return reinterpret_cast<const void*>(&k2_prune_v1<8, true>);
            case 4: return reinterpret_cast<const void*>(&k2_prune_v1<4, true>);
            case 2: return reinterpret_cast<const void*>(&k2_prune_v1<2, true>);
            case 1: return reinterpret_cast<const void*>(&k2_prune_v1<1, true>);
        }
        return nullptr;
    }
    switch (nf) {
        case 16: return reinterpret_cast<const void*>(&k2_prune_v1<16, false>);
        case 8: return reinterpret_cast<const void*>(&k2_prune_v1<8, false>);
        case 4: return reinterpret_cast<const void*>(&k2_prune_v1<4, false>);
        case 2: return reinterpret_cast<const void*>(&k2_prune_v1<2, false>);
        case 1: return reinterpret_cast<const void*>(&k2_prune_v1<1, false>);
    }
    return nullptr;
}
const void* k3_kernel(bool host_out) { return host_out ? reinterpret_cast<const void*>(&k3_score<true>) : reinterpret_cast<const void*>(&k3_score<false>); }
const void* k3x_kernel() { return reinterpret_cast<const void*>(&k3_score_x); }
const void* kx_collect_kernel() { return reinterpret_cast<const void*>(&k_x_collect); }
const void* gate_kernel() { return reinterpret_cast<const void*>(&k_gate); }
const void* kx_probe_kernel() { return reinterpret_cast<const void*>(&k_x_probe); }
const void* k3_cluster_kernel() { return reinterpret_cast<const void*>(&k3_cluster_score); }
const void* fetch_small_kernel() { return reinterpret_cast<const void*>(&k_fetch_small); }
const void* k4_kernel(int nf)
{
    switch (nf) {
        case 8: return reinterpret_cast<const void*>(&k4_viterbi<8>);
        case 4: return reinterpret_cast<const void*>(&k4_viterbi<4>);
        case 2: return reinterpret_cast<const void*>(&k4_viterbi<2>);
        case 1: return reinterpret_cast<const void*>(&k4_viterbi<1>);
    }
    return nullptr;
}

}  // namespace cafehip
