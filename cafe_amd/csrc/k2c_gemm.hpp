// k2c_gemm.hpp -- factor tables of compressed subtrees as a tiled GEMM (round 6; values of cafe/cafe_tree.c:213-271 with
// the mat-vec of libtree/birthdeath.c:163-182 recast as in k2_mfma.hpp).  No reference counterpart: the reference evaluates
// every family's subtree on its own.
//
// k2c_nodes (k2_mfma.hpp) gives a workgroup 16 states of one node: it gathers the states' whole node vectors into LDS, waits,
// then every wave streams ITS row tile of the node's matrix from L2 -- 4 flop per operand byte, the set-up of a tile (header,
// indices, gathers: half a tile's life on a large level, profiles/r05/k2c_stamps_paired_and_batched_gathers.txt) hidden only by
// whatever other tiles the CU holds.  Here a workgroup owns 16 * NST states of one node:
//   * every wave keeps NST state tiles x NRT_W row tiles of accumulators, so one matrix-operand load feeds NST matrix
//     instructions: 4 * NST flop per L2 byte, the matrix re-read NST times less often;
//   * the node vectors never exist as a whole: the K dimension is cut into chunks of KC k-steps (32 columns); chunk c + 1 of
//     L[state][k] = F_a[k] * F_b[k] is requested from the children's tables / matrix columns (16-byte loads into registers)
//     before the matrix instructions of chunk c start, multiplied and written to the other of two small LDS buffers
//     behind them, one barrier per chunk: after the first chunk the gathers' latency is covered by the workgroup's own
//     products, and the LDS a workgroup needs no longer grows with the states it owns (2 x MS x 272 bytes);
//   * the matrix operand's ring (k2_mfma.hpp: uniform base + fixed lane offsets, depth 4) runs across the chunk boundaries.
// Every accumulator still sees, k-step by k-step in ascending order, the same v_mfma_f64_16x16x4 on the same operands as in
// k2c_nodes and in the uncompressed walk: the table rows are BIT-IDENTICAL (tests/test_gpu_compression.py compares with ==).
#pragma once
#include "k2_mfma.hpp"

namespace {
using namespace cafehip;

constexpr int K2G_KC = 8;                 // k-steps per chunk
constexpr int K2G_CS = 4 * K2G_KC + 2;    // row stride of a chunk buffer in doubles: == 2 (mod 32), the A operand's reads are conflict-free
#ifndef CAFE_K2G_DEPTH
#define CAFE_K2G_DEPTH 4
#endif
constexpr int K2G_D = CAFE_K2G_DEPTH;     // depth of the matrix operand's ring (divides KC: ring slots are compile-time)

// The product of one workgroup's wave: NT (<= NRT_W) live row tiles.  Contains the chunk loop, i.e. the gathers and barriers
// every wave of the workgroup takes part in (the same number of barriers whatever NT).
template <int NST, int NRT_W, int NT, int GS>
__device__ __forceinline__ void k2g_product(const K2cArgs& a, double* Abuf, k2_gbytes sb, const unsigned (&voff)[NRT_W],
                                            const double* (&p0)[GS], const double* (&p1)[GS], const bool (&live)[GS],
                                            const int (&srow)[GS], int cidx, int li, int lk, cafe_d4 (&acc)[NST][NRT_W])
{
    constexpr int KC = K2G_KC, CS = K2G_CS, D = K2G_D, MS = 16 * NST;
    constexpr bool PAIR = NRT_W == 2 && NT == 2;
    static_assert(KC % D == 0, "ring slots must be compile-time across chunks");
    const int ksteps = a.ksteps;
    const int nchunks = (ksteps + KC - 1) / KC;
    const unsigned kstride_bytes = 32u * (unsigned)a.LD;

    cafe_d2 g0[GS], g1[GS];
    auto gather = [&](int cc) {   // request chunk cc of both children's rows (clamped rows: no branches)
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            g0[q] = *reinterpret_cast<const cafe_d2*>(p0[q] + cc * (4 * KC));
            g1[q] = *reinterpret_cast<const cafe_d2*>(p1[q] + cc * (4 * KC));
        }
    };
    auto deposit = [&](int cc) {   // L[state][k] = F_a[k] * F_b[k] -> chunk buffer cc & 1
        double* buf = Abuf + (cc & 1) * (MS * CS);
        const int k = cc * (4 * KC) + 2 * cidx;
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            if (srow[q] < MS) {
                cafe_d2 v;
                v.x = (live[q] && k < a.C) ? g0[q].x * g1[q].x : 0.0;
                v.y = (live[q] && k + 1 < a.C) ? g0[q].y * g1[q].y : 0.0;
                *reinterpret_cast<cafe_d2*>(buf + srow[q] * CS + 2 * cidx) = v;
            }
        }
    };

    double aq[2][NST], bq[D][NT];
    k2_gbytes bk = sb;
    unsigned vo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) vo[j] = k2_opaque(voff[j]);
#define K2G_LOADB(slot)                                                                                        \
    {                                                                                                          \
        if constexpr (PAIR) {                                                                                  \
            const cafe_d2 v2 = *(k2_gptr2)(bk + k2_opaque(vo[0]));                                             \
            bq[slot][0] = v2.x;                                                                                \
            bq[slot][1] = v2.y;                                                                                \
        } else {                                                                                               \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) bq[slot][j] = *(k2_gptr)(bk + k2_opaque(vo[j]));    \
        }                                                                                                      \
        bk += kstride_bytes;                                                                                   \
    }
#define K2G_LOADA(slot, kk)                                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NST; ++i) aq[slot][i] = pa[i * 16 * CS + (kk) * 4];              \
    }
#define K2G_MFMA(bslot, aslot)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NST; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                         \
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[aslot][i], bq[bslot][j], acc[i][j], 0, 0, 0);
    // region kk of a chunk: the matrix operand of k-step + D - 1 (it may run past the matrix's last k-step into the padding
    // behind the allocation's last slot -- loaded, never used), the node-vector operand of the chunk's next k-step, the
    // matrix instructions of this k-step
#define K2G_REGION(kk)                                                                                         \
    {                                                                                                          \
        K2G_LOADB(((kk) + D - 1) % D)                                                                          \
        if constexpr ((kk) + 1 < KC) K2G_LOADA(((kk) + 1) & 1, ((kk) + 1 < KC ? (kk) + 1 : 0))                 \
        K2G_MFMA((kk) % D, (kk) & 1)                                                                           \
        k2_region_pattern<(PAIR ? 1 : NT), ((kk) + 1 < KC ? NST : 0), NST * NT, true>();                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

    // prologue: chunk 0 lands in buffer 0, chunk 1 is in flight; the matrix operand's first D - 1 k-steps are requested
    gather(0);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) K2G_LOADB(d)
    deposit(0);
    gather(min(1, nchunks - 1));
    __syncthreads();
    // Whole chunks: ONE branch-free body -- the matrix operand's loads, the gathers and the waits for them come in a fixed
    // order, so the compiler's s_waitcnt vmcnt(N) are exact (vector-memory loads complete in order: with a branch anywhere in
    // the body it falls back to vmcnt(0) at every region, i.e. no ring at all -- disassembly of the first version).  Behind
    // the last chunk the body gathers the last chunk again and deposits it where nobody reads it.
    const int nfull = ksteps / KC;
    for (int c = 0; c < nfull; ++c) {
        const k2_lptr pa = (k2_lptr)(Abuf + (c & 1) * (MS * CS) + li * CS + lk);
        K2G_LOADA(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        K2G_REGION(0) K2G_REGION(1) K2G_REGION(2) K2G_REGION(3) K2G_REGION(4) K2G_REGION(5) K2G_REGION(6) K2G_REGION(7)
        deposit(c + 1);                          // (buffer (c + 1) & 1 was last read in iteration c - 1: a barrier ago)
        gather(min(c + 2, nchunks - 1));
        __syncthreads();
    }
    if (nfull < nchunks) {
        // the matrix's last, partial chunk (deposited by the last trip of the loop, or by the prologue): whole regions
        // skipped by wave-uniform branches
        const int c = nfull, kleft = ksteps - nfull * KC;
        const k2_lptr pa = (k2_lptr)(Abuf + (c & 1) * (MS * CS) + li * CS + lk);
        K2G_LOADA(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        K2G_REGION(0)
        if (kleft > 1) K2G_REGION(1)
        if (kleft > 2) K2G_REGION(2)
        if (kleft > 3) K2G_REGION(3)
        if (kleft > 4) K2G_REGION(4)
        if (kleft > 5) K2G_REGION(5)
        if (kleft > 6) K2G_REGION(6)
    }
#undef K2G_LOADB
#undef K2G_LOADA
#undef K2G_MFMA
#undef K2G_REGION
}

// NST state tiles per workgroup, NRT_W row tiles per wave (2: the even and the odd rows of 32, one 16-byte load per k-step),
// GS gather slots per thread (16 * NST states x 16 sixteen-byte columns per chunk over blockDim threads).
// MAXT: the largest workgroup the instantiation is launched with (512: 256 registers per lane, 1024: 128).
template <int NST, int NRT_W, int GS, int MAXT>
__global__ __launch_bounds__(MAXT) void k2c_gemm(K2cArgs a)
{
    static_assert(K2G_KC == 8, "the regions of a chunk are written out");
    constexpr int MS = 16 * NST, CS = K2G_CS;
    extern __shared__ double Abuf[];   // [2][MS][CS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    asm volatile("" ::"s"(a.PT), "s"(a.PTfold), "s"(a.node_key), "s"(a.n_nodes), "s"(a.tiles), "s"(a.leaf_has_err32), "s"(a.tables),
                 "s"(a.table_set_stride), "s"(a.C), "s"(a.LD), "s"(a.KP), "s"(a.LDv), "s"(a.ksteps), "s"(a.block_threads));
    K2C_STAMP(0);
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed, for speed only); XCD x takes a CONTIGUOUS eighth of the level's
    // tiles -- they are ordered by node -- so its L2 holds the matrices of an eighth of the level's nodes, not of all of them
    unsigned tile_ix = blockIdx.x;
    if (a.xcd_remap) {
        const unsigned nb = gridDim.x, x = blockIdx.x & 7u, per = nb >> 3, rem = nb & 7u;
        tile_ix = x * per + (x < rem ? x : rem) + (blockIdx.x >> 3);
    }
    const cafehip::CTile& t = a.tiles[tile_ix];
    const int set = blockIdx.y;
    const int T16 = a.block_threads >> 4;   // states a round of gather slots covers
    const int cidx = tid & 15;
    int srow[GS], i0[GS], i1[GS];
#pragma unroll
    for (int q = 0; q < GS; ++q) {
        srow[q] = (tid >> 4) + q * T16;
        const int sc = min(srow[q], MS - 1);
        i0[q] = t.idx[0][sc];
        i1[q] = t.idx[1][sc];
    }
    const int32_t* nk = a.node_key + set * a.n_nodes;
    const int32_t* lhe = a.leaf_has_err32 ? a.leaf_has_err32 : nk;   // (any readable words when there is no error model)
    const bool map_in_lanes = a.n_nodes <= 128;
    int keyv0 = 0, keyv1 = 0, flagv = 0;
    if (map_in_lanes) {
        keyv0 = nk[min(lane, a.n_nodes - 1)];
        keyv1 = nk[min(lane + 64, a.n_nodes - 1)];
        flagv = lhe[min(lane, (a.n_nodes + 1) / 2 - 1)];
    }
    int node = t.node, n_live = t.n_live, state0 = t.state0, out_off = t.out_off;
    int child0 = t.child[0], child1 = t.child[1], kind0 = t.kind[0], kind1 = t.kind[1];
    int leafcol0 = t.leafcol[0], leafcol1 = t.leafcol[1], tab_off0 = t.tab_off[0], tab_off1 = t.tab_off[1];
#define K2C_UNIFORM(x) x = __builtin_amdgcn_readfirstlane(x)
    K2C_UNIFORM(node); K2C_UNIFORM(n_live); K2C_UNIFORM(state0); K2C_UNIFORM(out_off); K2C_UNIFORM(child0); K2C_UNIFORM(child1);
    K2C_UNIFORM(kind0); K2C_UNIFORM(kind1); K2C_UNIFORM(leafcol0); K2C_UNIFORM(leafcol1); K2C_UNIFORM(tab_off0); K2C_UNIFORM(tab_off1);
    asm volatile("" : "+s"(node), "+s"(n_live), "+s"(state0), "+s"(out_off), "+s"(child0), "+s"(child1), "+s"(kind0), "+s"(kind1),
                 "+s"(leafcol0), "+s"(leafcol1), "+s"(tab_off0), "+s"(tab_off1));
    K2C_STAMP(1);
    double* const tab = a.tables + (size_t)set * a.table_set_stride;
    const int Wr = a.block_threads >> 6;
    const int RT = (a.C + 15) >> 4;
    const int rt_base = RT / Wr, rt_rem = RT - rt_base * Wr;
    const int ntile = rt_base + (wave < rt_rem ? 1 : 0);   // 1 .. NRT_W (the launcher gives every wave a tile)
    const int rt0 = wave * rt_base + min(wave, rt_rem);
    const bool leaf0 = kind0 == 0, leaf1 = kind1 == 0;
    int key_node, key_c0, key_c1, flag0, flag1;
    if (map_in_lanes) {
        auto pick = [&](int v) { return v < 64 ? __builtin_amdgcn_readlane(keyv0, v) : __builtin_amdgcn_readlane(keyv1, v - 64); };
        key_node = pick(node);
        key_c0 = pick(child0);
        key_c1 = pick(child1);
        flag0 = __builtin_amdgcn_readlane(flagv, leaf0 ? leafcol0 : 0);
        flag1 = __builtin_amdgcn_readlane(flagv, leaf1 ? leafcol1 : 0);
    } else {
        key_node = nk[node];
        key_c0 = nk[child0];
        key_c1 = nk[child1];
        flag0 = lhe[leaf0 ? leafcol0 : 0];
        flag1 = lhe[leaf1 ? leafcol1 : 0];
    }
    K2C_UNIFORM(key_node); K2C_UNIFORM(key_c0); K2C_UNIFORM(key_c1); K2C_UNIFORM(flag0); K2C_UNIFORM(flag1);
    asm volatile("" : "+s"(key_node), "+s"(key_c0), "+s"(key_c1), "+s"(flag0), "+s"(flag1));
#undef K2C_UNIFORM
    K2C_STAMP(2);
    const bool err0 = flag0 != 0, err1 = flag1 != 0;
    const size_t msz = (size_t)a.KP * a.LD;
    const k2_gbytes sb = k2_uniform(a.PT + (size_t)key_node * msz);
    const bool fold0 = a.PTfold != nullptr && a.leaf_has_err32 != nullptr && err0;
    const bool fold1 = a.PTfold != nullptr && a.leaf_has_err32 != nullptr && err1;
    const double* base0 = leaf0 ? (fold0 ? a.PTfold : a.PT) + (size_t)key_c0 * msz : tab + tab_off0;
    const double* base1 = leaf1 ? (fold1 ? a.PTfold : a.PT) + (size_t)key_c1 * msz : tab + tab_off1;
    const double *p0[GS], *p1[GS];
    bool live[GS];
#pragma unroll
    for (int q = 0; q < GS; ++q) {
        // a count beyond the column range: no such column (cafe/cafe_tree.c:208-209)
        live[q] = srow[q] < n_live && !(leaf0 && i0[q] > a.C - 1) && !(leaf1 && i1[q] > a.C - 1);
        p0[q] = base0 + (size_t)(live[q] ? i0[q] : 0) * a.LD + 2 * cidx;
        p1[q] = base1 + (size_t)(live[q] ? i1[q] : 0) * a.LD + 2 * cidx;
    }
    cafe_d4 fac[NST][NRT_W];
#pragma unroll
    for (int i = 0; i < NST; ++i)
#pragma unroll
        for (int j = 0; j < NRT_W; ++j) fac[i][j] = cafe_d4{0.0, 0.0, 0.0, 0.0};
    {
        unsigned voff[NRT_W];
#pragma unroll
        for (int j = 0; j < NRT_W; ++j) voff[j] = (unsigned)(lk * a.LD + li + ((j < ntile) ? (rt0 + j) : rt0) * 16) * 8u;
        if constexpr (NRT_W == 2) {
            if (ntile == 2) {
                voff[0] = (unsigned)(lk * a.LD + 2 * li + rt0 * 16) * 8u;   // rows rt0 * 16 + 2 li, + 1: one 16-byte load
                k2g_product<NST, NRT_W, 2, GS>(a, Abuf, sb, voff, p0, p1, live, srow, cidx, li, lk, fac);
            } else {   // (the last wave of an odd number of row tiles)
                k2g_product<NST, NRT_W, 1, GS>(a, Abuf, sb, voff, p0, p1, live, srow, cidx, li, lk, fac);
            }
        } else {
            k2g_product<NST, NRT_W, 1, GS>(a, Abuf, sb, voff, p0, p1, live, srow, cidx, li, lk, fac);
        }
    }
    K2C_STAMP(5);
    double* const out = tab + out_off + (size_t)state0 * a.LD;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = i * 16 + lk + 4 * r;
            if (f >= n_live) continue;
            if constexpr (NRT_W == 2) {
                if (ntile == 2) {
                    // the two rows of a pair side by side: one 16-byte store per lane (LD, the tables' offsets and the row are even)
                    const int row = rt0 * 16 + 2 * li;
                    cafe_d2 v2;
                    v2.x = (row < a.C) ? fac[i][0][r] : 0.0;
                    v2.y = (row + 1 < a.C) ? fac[i][1][r] : 0.0;
                    *reinterpret_cast<cafe_d2*>(out + (size_t)f * a.LD + row) = v2;
                    continue;
                }
            }
            const int row = rt0 * 16 + li;
            out[(size_t)f * a.LD + row] = (row < a.C) ? fac[i][0][r] : 0.0;
        }
    }
    K2C_STAMP(6);
}

}  // namespace
