// k1_matrices.hip -- K1: birth-death transition matrices for every unique (int branch length, lambda, mu) key of one
// evaluation == compute_birthdeath_rates, libtree/birthdeath.c:238-286; k1e_fold_error: the error model folded
// into those matrices (posterior path), cafe/cafe_tree.c:196-203.  gfx950 only.
#include "exp_like_host.hpp"
#include "kernels.hpp"
#include "k3_device.hpp"

namespace {
using namespace cafehip;

// Block (0,0,0) of the first launch of an evaluation mirrors the node -> matrix map from the pinned host block into
// device memory (the pruning launches index their matrices through it) and resets the first-zero-family slots the
// score kernel will atomicMin into.  n_sets / n_nodes come as arguments: ONE round trip to host memory, not a chain.
__device__ __forceinline__ void k1_mirror_node_keys(const K1Args& a, const bool first_block)
{
    if (!first_block) return;
    if (a.node_key_dev) {
        const int32_t* __restrict__ src = eval_node_key(a.ep, a.key_cap);
        for (int i = threadIdx.x; i < a.n_sets * a.n_nodes; i += 256) {
            const int set = i / a.n_nodes;
            a.node_key_dev[(size_t)a.set_row[set] * a.n_nodes + (i - set * a.n_nodes)] = src[i];
        }
    }
    if (a.first_zero && (int)threadIdx.x < a.n_sets) a.first_zero[threadIdx.x] = INT32_MAX;
    if (a.n_prior > 0) {
        // a new prior (once per search): mirrored by the same kernel stores as the map, so that it is ordered with
        // the launches that read it like everything else of the evaluation (a copy command is not: round 3)
        const double* __restrict__ pr = eval_prior(a.ep, a.prior_offset);
        for (int i = threadIdx.x; i < a.n_prior; i += 256) {
            a.prior_dev[i] = pr[i];
            a.logprior_dev[i] = pr[kMaxPrior + i];
        }
    }
}

// ------------------------------------------------------------------------------------
// K1: transition matrices.  One 16x16 tile of (s, c) entries per workgroup; the 16
// consecutive s of a tile are the fast lane index so that the transposed store
// PT[c][s] is 128 B contiguous.  The two ln C runs of each of the 16 rows are staged
// in LDS (odd row stride -> conflict-free ds_read_b64).
// Arithmetic follows libtree/birthdeath.c:52-73 / :34-50 term by term, j ascending,
// running product for coeff^j, clamp to [0,1]; contraction is off so each term is the
// same sequence of IEEE operations as the reference's x86-64 build.
// ------------------------------------------------------------------------------------
// exp() of a term in the exact form: the host libm's own operation sequence where it was recognised (exp_like_host.hpp), so
// that the matrices of the report phase carry the bits the reference's build would produce on this machine
__device__ __forceinline__ double k1_exp(double t, int variant)
{
    if (variant == 1) return exp_like_host<true>(t);
    if (variant == 2) return exp_like_host<false>(t);
    return exp(t);
}

#pragma clang fp contract(off)
template <bool USE_LDS, bool PRODUCT_FORM>
__global__ __launch_bounds__(256) void k1_build_matrices(K1Args ka)
{
    const int exp_variant = ka.exp_variant;
    // `ep` is this evaluation's parameter block in PINNED HOST memory (read over the fabric: one 80-byte
    // KeyParam per workgroup); block (0,0,0) mirrors the node -> key map into device memory for the later
    // launches, so an evaluation needs no separate host-to-device copy.
    const EvalHeader* __restrict__ ep = ka.ep;
    const KeyParam* __restrict__ keys = eval_keys(ep);
    const int nkeys = ka.nkeys, keys_per_block = ka.keys_per_block, M = ka.M, LD = ka.LD, KP = ka.KP, ld_lnc = ka.ld_lnc;
    double* __restrict__ PT = ka.PT;
    const double* __restrict__ lncA = ka.tabA;
    const double* __restrict__ lncB = ka.tabB;
    extern __shared__ double k1_smem[];
    // issue the (slow, host-memory) read of this block's first key before the table staging so that the
    // two latencies overlap
    const KeyParam kp_first = keys[min((int)blockIdx.z * keys_per_block, nkeys - 1)];
    k1_mirror_node_keys(ka, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0);
    const int s0 = blockIdx.y * 16;
    const int c0 = blockIdx.x * 16;
    const int tx = threadIdx.x & 15;  // s within tile
    const int ty = threadIdx.x >> 4;  // c within tile
    const int s = s0 + tx;
    const int c = c0 + ty;

    // The table runs of this (s, c) tile do not depend on the key: stage them ONCE and build the tile
    // for keys_per_block keys.  A needs j <= min(s, c) <= min(s0, c0) + 15; B needs i = c - j <= c0 + 15.
    const double* a;
    const double* b;
    if (USE_LDS) {
        const int nA = min(min(s0, c0) + 16, M + 1);
        const int nB = min(c0 + 16, M + 1);
        double* sA = k1_smem;                        // [16][ld_lnc]
        double* sB = k1_smem + 16 * (size_t)ld_lnc;  // [16][ld_lnc]
        // thread (r = tid / 16, l = tid % 16) copies row s0 + r, columns l, l+16, ...: 128-byte runs,
        // eight loads in flight per thread before the first LDS store (the copy is latency-bound)
        {
            const int r = threadIdx.x >> 4, l = threadIdx.x & 15;
            const int sr = min(s0 + r, M);
            const double* ga = lncA + (size_t)sr * ld_lnc;
            const double* gb = lncB + (size_t)sr * ld_lnc;
            double* da = sA + r * ld_lnc;
            double* db = sB + r * ld_lnc;
            for (int i0 = l; i0 < nA; i0 += 128) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (i0 + 16 * u < nA) ? ga[i0 + 16 * u] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 16 * u < nA) da[i0 + 16 * u] = v[u];
            }
            for (int i0 = l; i0 < nB; i0 += 128) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (i0 + 16 * u < nB) ? gb[i0 + 16 * u] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 16 * u < nB) db[i0 + 16 * u] = v[u];
            }
        }
        __syncthreads();
        a = sA + tx * ld_lnc;
        b = sB + tx * ld_lnc;
    } else {
        a = lncA + (size_t)min(s, M) * ld_lnc;
        b = lncB + (size_t)min(s, M) * ld_lnc;
    }
    if (s > M || c > M) return;
    const int m = min(s, c);
    const int key_end = min(nkeys, (int)(blockIdx.z + 1) * keys_per_block);
    for (int key = blockIdx.z * keys_per_block; key < key_end; ++key) {
        const KeyParam kp = (key == (int)blockIdx.z * keys_per_block) ? kp_first : keys[key];
        double p;
        if (s == 0) {
            p = (c == 0) ? 1.0 : 0.0;  // row 0 is e_0 in every mode (libtree/birthdeath.c:244, :212-215)
        } else if (kp.mode < 2) {
            p = (kp.mode == 1 && s == c) ? 1.0 : 0.0;  // zero / identity matrices
        } else {
            p = 0.0;
            // terms are accumulated strictly in j order (the reference's order)
            if (PRODUCT_FORM && kp.fast_ok) {
                // Same sum with the exponentials factored: a/b hold exp(ln C) (binomials through the
                // reference's Lanczos lgamma), the power part alpha^(..) coeff^j is a geometric sequence
                // kept as mantissa in [1,2) x 2^e so that nothing under/overflows before the final ldexp.
                // One exp2 per ENTRY instead of one exp per TERM; deviation from the per-term form
                // <~ 5e-13 relative, the size of the rounding the reference itself commits forming t.
                const double y0 = (kp.mode == 2) ? (double)(s + c) * kp.l2a : (double)s * kp.l2a + (double)c * kp.l2b;
                const double e0 = floor(y0);
                double gm = exp2(y0 - e0);
                int e = (int)e0;
#pragma unroll 4
                for (int j = 0; j <= m; ++j) {
                    const double term = a[j] * b[c - j] * gm;
                    p += ldexp(term, e);
                    gm *= kp.rho_m;
                    e += kp.rho_e;
                    if (gm >= 2.0) {
                        gm *= 0.5;
                        e += 1;
                    }
                }
            } else if (kp.mode == 2) {
                double lastterm = 1.0;
                const int s_add_c = s + c;
#pragma unroll 4
                for (int j = 0; j <= m; ++j) {
                    const double t = a[j] + b[c - j] + (double)(s_add_c - 2 * j) * kp.log_alpha;
                    p += k1_exp(t, exp_variant) * lastterm;
                    lastterm *= kp.coeff;
                }
            } else {
#pragma unroll 4
                for (int j = 0; j <= m; ++j) {
                    const double t = a[j] + b[c - j] + (double)(s - j) * kp.log_alpha +
                                     (double)(c - j) * kp.log_beta + (double)j * kp.log_coeff;
                    p += k1_exp(t, exp_variant);
                }
            }
            p = fmax(fmin(p, 1.0), 0.0);  // MAX(MIN(p,1),0)
        }
        PT[(size_t)kp.slot * KP * LD + (size_t)c * LD + s] = p;
    }
}

// ------------------------------------------------------------------------------------
// K1, register-blocked product form.  Thread = one row s x K1Q consecutive columns c..c+K1Q-1.
// For a fixed row the power part of term j is the same geometric sequence for every column up to a
// per-column constant (alpha^q or beta^q), so the K1Q sums share a[j] * rho^j and slide a window over
// the second binomial run b[c + q - j]:  per 8 terms of K1Q entries the thread issues 8 + 8 LDS reads
// and 8 + 8 + 8*K1Q FP64 operations, instead of 2 reads + ~10 operations per single term.
// rho^j is carried as (g in [1,2)) * 2^e, renormalised once per 8-term chunk; inside a chunk plain
// doubles are safe because the host marks a key fast_ok == 2 only if binomials * rho^8 < 2^1000.
// Terms are still accumulated in the reference's order (j ascending).  Negative b indices (j > c + q)
// and a[j] beyond s read staged zeros, which add exact zeros.
// ------------------------------------------------------------------------------------
#ifndef CAFEHIP_K1Q
#define CAFEHIP_K1Q 8
#endif
constexpr int K1Q = CAFEHIP_K1Q;
constexpr int K1_BPAD = 24;  // zeros in front of every staged B row (window indices down to -22)

// (the kernel's body as a function of the block's position, so that it can also run as the trailing blocks of the score
// kernel's launch: k3_score_then_k1_rb below)
__device__ __forceinline__ void k1_rb_block(const K1Args& ka, const int bx, const int by, const int bz)
{
    // `ep` is this evaluation's parameter block in PINNED HOST memory (read over the fabric: one 80-byte
    // KeyParam per workgroup); block (0,0,0) mirrors the node -> key map into device memory for the later
    // launches, so an evaluation needs no separate host-to-device copy.
    const EvalHeader* __restrict__ ep = ka.ep;
    const KeyParam* __restrict__ keys = eval_keys(ep);
    const int nkeys = ka.nkeys, keys_per_block = ka.keys_per_block, M = ka.M, LD = ka.LD, KP = ka.KP, ld_lnc = ka.ld_lnc;
    double* __restrict__ PT = ka.PT;
    const double* __restrict__ expA = ka.tabA;
    const double* __restrict__ expB = ka.tabB;
    extern __shared__ double k1_smem[];
    const KeyParam kp_first = keys[min(bz * keys_per_block, nkeys - 1)];
    k1_mirror_node_keys(ka, bx == 0 && by == 0 && bz == 0);
    const int s0 = by * 16;
    const int c0 = bx * (16 * K1Q);
    const int tx = threadIdx.x & 15;   // row within the tile (fast lane index: PT[c][s] stores are 128-byte runs)
    const int tq = threadIdx.x >> 4;   // column group
    const int s = s0 + tx;
    const int cb = c0 + tq * K1Q;      // first column of this thread

    // staged runs: A needs j <= min(s, c+q) (+7 chunk overrun), B needs i = c + q - j in [-22, c0 + 16*K1Q)
    const int ldA = ld_lnc + 8;                 // odd + 8 = odd: conflict-free over the 16 rows
    const int ldB = ld_lnc + K1_BPAD + 8;       // odd
    const int nA = min(min(s0 + 16, c0 + 16 * K1Q), M + 1) + 8;
    const int nB = min(c0 + 16 * K1Q, M + 1) + 8;
    double* sA = k1_smem;
    double* sB = k1_smem + 16 * (size_t)ldA;
    {
        const int r = threadIdx.x >> 4, l = threadIdx.x & 15;
        const int sr = min(s0 + r, M);
        const double* ga = expA + (size_t)sr * ld_lnc;
        const double* gb = expB + (size_t)sr * ld_lnc;
        double* da = sA + r * ldA;
        double* db = sB + r * ldB + K1_BPAD;
        for (int i0 = l; i0 < nA; i0 += 128) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i0 + 16 * u <= M) ? ga[i0 + 16 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 16 * u < nA) da[i0 + 16 * u] = v[u];
        }
        for (int i0 = l; i0 < nB; i0 += 128) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i0 + 16 * u <= M) ? gb[i0 + 16 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 16 * u < nB) db[i0 + 16 * u] = v[u];
        }
        // (requesting both runs of a round together -- half the round trips -- measured SLOWER in round 3: K1 17.2 -> 19.2 us
        // at configs[1], 50 -> 55 us at configs[2]; profiles/r03/k1_k3_k2c_micro_sweeps.txt)
        for (int i = l; i < K1_BPAD; i += 16) sB[r * ldB + i] = 0.0;
    }
    __syncthreads();
    if (s > M || cb > M) return;
    const double* a = sA + tx * ldA;
    const double* b = sB + tx * ldB + K1_BPAD;
    const int mmax = min(s, min(cb + K1Q - 1, M));
    const int key_end = min(nkeys, (bz + 1) * keys_per_block);
    for (int key = bz * keys_per_block; key < key_end; ++key) {
        const KeyParam kp = (key == bz * keys_per_block) ? kp_first : keys[key];
        double p[K1Q];
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < K1Q; ++q) p[q] = (cb + q == 0) ? 1.0 : 0.0;  // row 0 is e_0 in every mode
        } else if (kp.mode < 2) {
#pragma unroll
            for (int q = 0; q < K1Q; ++q) p[q] = (kp.mode == 1 && s == cb + q) ? 1.0 : 0.0;  // zero / identity
        } else if (kp.fast_ok == 2) {
            double gm0[K1Q];
            int e0[K1Q];
#pragma unroll
            for (int q = 0; q < K1Q; ++q) {
                const double y0 = (kp.mode == 2) ? (double)(s + cb + q) * kp.l2a
                                                 : (double)s * kp.l2a + (double)(cb + q) * kp.l2b;
                const double ef = floor(y0);
                gm0[q] = exp2(y0 - ef);
                e0[q] = (int)ef;
                p[q] = 0.0;
            }
            const double rho = ldexp(kp.rho_m, kp.rho_e);
            double g = 1.0;
            int e = 0;
            double win[K1Q + 7];  // win[d + 7] = b[cb - j0 + d], d in [-7, K1Q)
#pragma unroll
            for (int t = 0; t < K1Q + 7; ++t) win[t] = b[cb - 7 + t];
            for (int j0 = 0; j0 <= mmax; j0 += 8) {
                double acc[K1Q];
#pragma unroll
                for (int q = 0; q < K1Q; ++q) acc[q] = 0.0;
                double gu = g;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double ag = a[j0 + u] * gu;
#pragma unroll
                    for (int q = 0; q < K1Q; ++q) acc[q] = fma(ag, win[q - u + 7], acc[q]);
                    gu *= rho;
                }
#pragma unroll
                for (int q = 0; q < K1Q; ++q) p[q] += ldexp(acc[q], e + e0[q]);
                int ex;
                g = 2.0 * frexp(gu, &ex);  // gu = g * 2^(ex - 1), g in [1, 2)
                e += ex - 1;
                // slide the window by 8 terms: indices move down by 8
#pragma unroll
                for (int t = K1Q + 6; t >= 8; --t) win[t] = win[t - 8];
#pragma unroll
                for (int t = 0; t < 8 && t < K1Q + 7; ++t) win[t] = b[cb - j0 - 15 + t];
            }
#pragma unroll
            for (int q = 0; q < K1Q; ++q) p[q] = fmax(fmin(p[q] * gm0[q], 1.0), 0.0);
        } else {
            // keys whose rho^8 could leave the double range: per-term mantissa/exponent form (as k1_build_matrices)
#pragma unroll 1
            for (int q = 0; q < K1Q; ++q) {
                const int c = cb + q;
                if (c > M) {
                    p[q] = 0.0;
                    continue;
                }
                const int m = min(s, c);
                const double y0 = (kp.mode == 2) ? (double)(s + c) * kp.l2a : (double)s * kp.l2a + (double)c * kp.l2b;
                const double e0 = floor(y0);
                double gm = exp2(y0 - e0);
                int e = (int)e0;
                double acc = 0.0;
                for (int j = 0; j <= m; ++j) {
                    const double term = a[j] * b[c - j] * gm;
                    acc += ldexp(term, e);
                    gm *= kp.rho_m;
                    e += kp.rho_e;
                    if (gm >= 2.0) {
                        gm *= 0.5;
                        e += 1;
                    }
                }
                p[q] = fmax(fmin(acc, 1.0), 0.0);
            }
        }
#pragma unroll
        for (int q = 0; q < K1Q; ++q)
            if (cb + q <= M) PT[(size_t)kp.slot * KP * LD + (size_t)(cb + q) * LD + s] = p[q];
    }
}

// Linear workgroup index -> (column block, row block, key block).  The terms of entry (s, c) number min(s, c) + 1, so the tiles
// of one key differ in work by an order of magnitude.  Where a launch is several rounds of workgroups (3+ per CU: 40-62 keys of
// a 151- or 251-wide matrix) the (tile, key) pairs are dealt HEAVIEST FIRST, so that the launch does not end on a heavy tile
// that started late: -11 ... -15 % (configs[2] 52.2 -> 44.1 us, the configs[3] shard 34.7 -> 30.9 us).  A launch whose
// workgroups are all resident from the start (configs[1]: 500 on 256 CUs) is NOT helped by any order -- heavy half / light half,
// alternating, heaviest first all measured 5-9 % slower than the grid order there: it is a chain of latencies (key, staging,
// stores), not of arithmetic.  A matter of speed only: every tile is built by exactly one workgroup whatever the placement.
__device__ __forceinline__ void k1_rb_coords(const K1Args& ka, const int lin, int& bx, int& by, int& bz)
{
    if (!ka.balanced) {
        bx = lin % ka.gx;
        const int t = lin / ka.gx;
        by = t % ka.gy;
        bz = t / ka.gy;
        return;
    }
    const int total = ka.gx * ka.gy * ka.gz, half = (total + 1) >> 1;
    // balanced: 1 = heavy half then light half ascending (k and k + half meet), 2 = heavy / light alternating (neighbours meet),
    // 3 = heaviest first throughout
    const int r = ka.balanced == 1 ? (lin < half ? lin : total - 1 - (lin - half))
                : ka.balanced == 2 ? ((lin & 1) ? total - 1 - (lin >> 1) : (lin >> 1))
                                   : lin;
    const int tile = ka.tile_of_rank[r / ka.gz];
    bz = r % ka.gz;
    bx = tile % ka.gx;
    by = tile / ka.gx;
}

__global__ __launch_bounds__(256) void k1_build_matrices_rb(K1Args ka)
{
    int bx, by, bz;
    k1_rb_coords(ka, blockIdx.x, bx, by, bz);
    k1_rb_block(ka, bx, by, bz);
}

// Score kernel and the matrices of the NEXT candidates in one launch (round 5).  Blocks [0, k3_blocks) are k3_score<true>'s
// (dispatched first: the score reaches the host as early as from its own launch); the blocks behind them build the
// matrices of the parameter sets the optimiser may ask for next, into cache entries -- in the time the chip otherwise
// idles: the tail of the score kernel and the host's turn-around (result pick-up, the optimiser's decision, the next
// launch: ~10-15 us).  The next evaluation's launches queue behind this one on the same stream, so no event, no second
// queue and no contention with the walk are involved; built on a second stream beside the walk, the same work cost the
// walk 6 us of its 55 (profiles/r05).
__global__ __launch_bounds__(256) void k3_score_then_k1_rb(K3K1Args a)
{
    if ((int)blockIdx.x < a.k3_blocks) {
        k3_score_block<true>(a.k3, blockIdx.x, a.k3_blocks, 0, 1);
        return;
    }
    int bx, by, bz;
    k1_rb_coords(a.k1, (int)blockIdx.x - a.k3_blocks, bx, by, bz);
    k1_rb_block(a.k1, bx, by, bz);
}
#pragma clang fp contract(fast)

// Error model folded into the matrices (posterior mode).  For a leaf with an error model the edge factor of a
// family is  sum_k errormatrix[observed][k] * P[row][k]  (cafe/cafe_tree.c:196-203 then :213-224), a function of
// (matrix, observed count, row) only -- not of the family.  It is formed once per evaluation,
//   PTfold[key][observed][row] = sum_{k ascending} err[observed][k] * PT[key][k][row],
// in the same order as the per-family sums of the walk, and the leaf becomes a plain column gather on PTfold.
// Not usable with per-row column limits (batch mode clips the sum at col_max of each row).
__global__ __launch_bounds__(256) void k1e_fold_error(FoldArgs a)
{
    const int key = blockIdx.z, cnt = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.LD) return;
    const double* P = a.PT + (size_t)key * a.KP * a.LD + s;
    const double* erow = a.err + (size_t)cnt * a.err_ld;
    const int klo = a.banded ? max(cnt + a.dlo, 0) : 0;
    const int khi = a.banded ? min(cnt + a.dhi, a.C - 1) : a.C - 1;
    double v = 0.0;
    for (int k = klo; k <= khi; ++k) v += erow[k] * P[(size_t)k * a.LD];
    a.PTfold[(size_t)key * a.KP * a.LD + (size_t)cnt * a.LD + s] = v;
}

}  // namespace

namespace cafehip {

const void* k1_kernel(bool use_lds, bool product_form)
{
    if (use_lds) return product_form ? reinterpret_cast<const void*>(&k1_build_matrices<true, true>) : reinterpret_cast<const void*>(&k1_build_matrices<true, false>);
    return product_form ? reinterpret_cast<const void*>(&k1_build_matrices<false, true>) : reinterpret_cast<const void*>(&k1_build_matrices<false, false>);
}
const void* k1_rb_kernel() { return reinterpret_cast<const void*>(&k1_build_matrices_rb); }
const void* k3_then_k1_rb_kernel() { return reinterpret_cast<const void*>(&k3_score_then_k1_rb); }
int k1_rb_columns() { return K1Q; }
int k1_rb_bpad() { return K1_BPAD; }
const void* k1e_fold_kernel() { return reinterpret_cast<const void*>(&k1e_fold_error); }

}  // namespace cafehip
