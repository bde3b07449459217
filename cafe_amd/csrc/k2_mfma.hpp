// k2_mfma.hpp -- K2 on the gfx950 matrix cores (included by cafehip.hip).
//
// Because the transition matrix of an edge is shared by every family
// (libtree/birthdeath.h:26-31), the per-family mat-vecs of one edge
// (cafe/cafe_tree.c:213-224) are the FP64 GEMM
//        Y[fam][row] = sum_k L[fam][k] * PT[k][row]
// issued as v_mfma_f64_16x16x4_f64 tiles: A = 16 families x 4 k from the LDS node
// buffer, B = 4 k x 16 rows straight from the transposed matrix in L2 (4 x 128 B
// segments per wave load), D = 16 families x 16 rows in registers.
//
// One workgroup owns NF = 16*NFT families and walks the whole tree for them:
//   * ONE node-vector buffer Lbuf[NF][LDv] in LDS; results that are not consumed by the
//     next step are parked (MfmaSchedule) in further LDS buffers or in a slot of a global scratch
//     that the workgroup owns while it runs (k2_acquire_park_slot);
//   * waves are arranged Wf x Wr: wave (wf, wr) owns family tiles [wf*NFT_W, +NFT_W) and an even
//     share (<= NRT_W) of the step's row tiles: NFT_W*NRT_W accumulator tiles (4 f64 per lane each);
//   * leaf children are column gathers PT[count][row] in the D layout; the Hadamard
//     product of the two child factors is taken in registers (cafe/cafe_tree.c:261-266).
// Layouts (lane l of a wave): A lane holds L[fam0 + (l&15)][k0 + (l>>4)];
// B lane holds PT[k0 + (l>>4)][row0 + (l&15)]; D reg r holds
// Y[fam0 + (l>>4) + 4r][row0 + (l&15)]   (verified by tools/mfma_f64_probe.hip).
#pragma once
#include <climits>
#include <cmath>

#include "kernels.hpp"

namespace {
using namespace cafehip;

typedef double cafe_d4 __attribute__((ext_vector_type(4)));


// Phase timeline of the walk for tools/k2_stamps.py: lane 0 of every wave records the shader clock at fixed points
// (slot 0 kernel start, 1 after the prologue, 2 + 6 * step + {0: leaf gathers issued, 1: first child factor done,
// 2: second child factor done, 3: past the read barrier, 4: result written + visible}, last: after the epilogue).
// Compiled out of the product library.
#ifdef CAFE_K2_STAMPS
#define K2_STAMP(slot)                                                                                         \
    do {                                                                                                       \
        if (a.stamps && lane == 0 && (slot) < K2_STAMP_SLOTS)                                                  \
            a.stamps[((size_t)blockIdx.x * 8 + wave) * K2_STAMP_SLOTS + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define K2_STAMP(slot) ((void)0)
#endif
#ifdef CAFE_K2_STAMPS
#define K2C_STAMP(slot)                                                                                        \
    do {                                                                                                       \
        if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 16 + wave) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define K2C_STAMP(slot) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------
// One edge product: acc[i][j] += node-vector tile(i) x matrix tile(j) over all k-steps, explicitly software-
// pipelined.  Left to itself hipcc schedules a two-stage source loop into
//     [3 B loads for k+1] [15 MFMAs of k] [3 B loads for k+2, 5 ds_read2 for k+1 AND k+2] wait lgkmcnt [15 MFMAs of k+1]
// i.e. the LDS latency of the node-vector operand is exposed once per two k-steps and the matrix operand has ONE
// k-step of MFMAs (240 cycles alone, ~400 shared with the SIMD's other wave) to cover an L2 round trip of ~500
// cycles (disassembly of k2_prune_mfma4<5,3>, round 2).  Here both operands live in rings of D
// k-steps (D = CAFE_K2_DEPTH16 / CAFE_K2_DEPTH4): region k (fenced by sched_barrier, so no load can sink below or hoist above it) issues the loads of
// k-step k + D - 1 and the matrix instructions of k-step k, so every operand has D - 1 whole regions to arrive
// wherever the in-region scheduler puts its load.  Same instructions on the same operands in the same order per
// accumulator as the plain loop: bit-identical results.
// ---------------------------------------------------------------------------------------------------
// Depth: the 16x16x4 kernel's regions are long (64 cycles per matrix instruction, 3-4 waves per SIMD), one region
// of lookahead covers the L2 round trip and the smaller rings keep a fourth wave resident (cfg 3: 6.63 -> 6.53 ms);
// the 4x4x4 kernel (16-cycle instructions, 2 waves per SIMD) needs two (cfg 2: 0.180 -> 0.164 ms).
#ifndef CAFE_K2_DEPTH16
#define CAFE_K2_DEPTH16 2
#endif
#ifndef CAFE_K2_DEPTH4
#define CAFE_K2_DEPTH4 3
#endif
static_assert(CAFE_K2_DEPTH16 >= 2 && CAFE_K2_DEPTH4 >= 2, "the operand rings need at least two slots");
#ifndef CAFE_K2_INTERLEAVE
#define CAFE_K2_INTERLEAVE 1
#endif

// inside a region: one operand load, then its share of the matrix instructions, ... so that the loads issue in
// the shadow of this wave's own MFMAs instead of ahead of them (sched_group_barrier: 0x20 VMEM read, 0x100 DS
// read, 0x8 MFMA).  DS_FIRST (the 16x16 kernel, depth 2): the node-vector (LDS) reads lead, because every matrix
// instruction of the NEXT region needs them while a matrix-operand load feeds one column of instructions -- and a
// read placed after the region's last matrix instruction lets the register allocator fold the two ring slots into
// one register, i.e. wait for the LDS round trip at the top of every region (disassembly).  The 4x4 kernel (depth
// 3, two regions of slack either way) measures 4 % faster with the global loads leading (cfg 2: 0.164 vs 0.171 ms).
template <int N_VMEM, int N_DS, int N_MFMA, bool DS_FIRST>
__device__ __forceinline__ void k2_region_pattern()
{
#if CAFE_K2_INTERLEAVE
    constexpr int n_loads = N_VMEM + N_DS;
    constexpr int q = N_MFMA / n_loads > 0 ? N_MFMA / n_loads : 1;
    constexpr int n_paired = N_MFMA < n_loads ? N_MFMA : n_loads;   // loads that get a matrix instruction behind them
    constexpr int n_first = DS_FIRST ? N_DS : N_VMEM;
#pragma unroll
    for (int i = 0; i < n_loads; ++i) {
        if ((i < n_first) == DS_FIRST) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        if (i < n_paired) __builtin_amdgcn_sched_group_barrier(0x8, q, 0);
    }
    if constexpr (N_MFMA - q * n_paired > 0) __builtin_amdgcn_sched_group_barrier(0x8, N_MFMA - q * n_paired, 0);
#endif
}

// The matrix operand's address is split into a WAVE-UNIFORM base (scalar registers: matrix + k-step row block,
// advanced by one scalar add per region) and per-lane 32-bit byte offsets that do not change during a product
// (lane's k row and matrix row + the wave's row tiles), i.e. `global_load_dwordx2 v, v_off, s[base:base+1]`;
// the node-vector operand uses one LDS address per family group, bumped once per D regions, with the k-step as
// the instruction's immediate offset.  A region is then [NT global loads, G (NFT_W) LDS reads, 2 scalar adds,
// the matrix instructions]: the 64-bit multiply-adds per load that a `base[k * stride + off]` form costs (10
// scalar + 9 vector instructions per region, disassembly of round 2's first version) competed with the matrix
// instructions for the wave's single issue slot per cycle.  The last D - 1 regions have no loads at all.
// (explicit address spaces: a pointer rebuilt from integers, or carried through the loop in an array, would
// otherwise degrade every operand load to a flat_load)
typedef const double __attribute__((address_space(1))) * k2_gptr;
typedef const char __attribute__((address_space(1))) * k2_gbytes;
typedef const double __attribute__((address_space(3))) * k2_lptr;
// two adjacent matrix rows in one 16-byte load (PAIR products: a wave's two row tiles are the even and the odd rows of 32)
typedef double cafe_d2 __attribute__((ext_vector_type(2)));
typedef const cafe_d2 __attribute__((address_space(1))) * k2_gptr2;

// keeps the 32-bit lane offset a 32-bit value at the load (hoisted out of the loop and widened there, it would
// cost a 64-bit vector add per load instead of the scalar-base addressing mode)
__device__ __forceinline__ unsigned k2_opaque(unsigned v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ k2_gbytes k2_uniform(const double* p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (k2_gbytes)(((unsigned long long)hi << 32) | lo);
}

// The k loop runs D regions per trip with compile-time ring slots.  ksteps - (D - 1) regions carry loads; their
// count modulo D is absorbed by a partial FIRST trip (phase p: regions p..D-1 only, the rings filled from slot p),
// so that the loop always ends on a trip boundary and the D - 1 load-free regions that drain the rings have fixed
// slots.  NT <= NRT_W: the wave's live row tiles; columns NT.. of acc are left untouched (a wave that was dealt one
// tile fewer than the widest must not burn matrix-pipe cycles on a dummy column: the SIMD's other waves can use them).
#define CAFE_K2_EDGE_BODY(NA, MFMA_BUILTIN)                                                                    \
    double aq[D][NA], bq[D][NT];                                                                               \
    k2_gbytes bk = sb; /* row block of the next k-step to load (wave-uniform) */                               \
    unsigned vo[NT];                                                                                           \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) vo[j] = k2_opaque(voff[j]);                                 \
    if (ksteps < D - 1) { /* (matrix side <= 4: not a real table, kept correct) */                             \
        for (int k = 0; k < ksteps; ++k) {                                                                     \
            CAFE_K2_LOAD(0, k)                                                                                 \
            CAFE_K2_MFMA(0)                                                                                    \
        }                                                                                                      \
        return;                                                                                                \
    }                                                                                                          \
    const int n_main = ksteps - (D - 1);                                                                       \
    const int full = n_main / D, rem = n_main - full * D;                                                      \
    const int p = rem ? D - rem : 0;                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) pa[i] -= 4 * p;                                             \
    _Pragma("unroll") for (int pp = 0; pp < D; ++pp)                                                           \
        if (p == pp) {                                                                                         \
            _Pragma("unroll") for (int d = 0; d < D - 1; ++d) CAFE_K2_LOAD((pp + d) % D, pp + d)               \
        }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (p > 0) {                                                                                               \
        _Pragma("unroll") for (int u = 1; u < D; ++u)                                                          \
            if (u >= p) CAFE_K2_REGION_L(u)                                                                    \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) pa[i] += 4 * D;                                         \
    }                                                                                                          \
    for (int it = 0; it < full; ++it) {                                                                        \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) vo[j] = k2_opaque(voff[j]);                             \
        _Pragma("unroll") for (int u = 0; u < D; ++u) CAFE_K2_REGION_L(u)                                      \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) pa[i] += 4 * D;                                         \
    }                                                                                                          \
    _Pragma("unroll") for (int u = 0; u < D - 1; ++u) CAFE_K2_REGION_N(u)

#define CAFE_K2_LOAD(slot, koff)                                                                               \
    {                                                                                                          \
        if constexpr (PAIR_) {                                                                                 \
            _Pragma("unroll") for (int j = 0; j < NT / 2; ++j) {                                               \
                const cafe_d2 v2 = *(k2_gptr2)(bk + vo[j]);                                                    \
                bq[slot][2 * j] = v2.x;                                                                        \
                bq[slot][2 * j + 1] = v2.y;                                                                    \
            }                                                                                                  \
        } else {                                                                                               \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) bq[slot][j] = *(k2_gptr)(bk + vo[j]);               \
        }                                                                                                      \
        bk += kstride_bytes;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NA_; ++i) aq[slot][i] = pa[i][(koff) * 4];                       \
    }
#define CAFE_K2_REGION_L(u)                                                                                    \
    {                                                                                                          \
        CAFE_K2_LOAD(((u) + D - 1) % D, (u) + D - 1)                                                           \
        CAFE_K2_MFMA((u) % D)                                                                                  \
        k2_region_pattern<(PAIR_ ? NT / 2 : NT), NA_, NA_ * NT, DS_FIRST_>();                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
#define CAFE_K2_REGION_N(u)                                                                                    \
    {                                                                                                          \
        CAFE_K2_MFMA((u) % D)                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

// PAIR (k2c_nodes on levels of many tiles, round 5): the NT (even) row tiles of the wave are read as NT / 2 16-byte loads, lane
// li of a pair holding rows 2 li and 2 li + 1 of the pair's 32: voff[0 .. NT/2) are the pairs' lane offsets, accumulator
// column 2 q carries the even rows of pair q and column 2 q + 1 the odd rows.  Half the vector-memory instructions for the
// same bytes; every accumulator still sees the same operands in the same k order.  (The family walk with the same dealing,
// gathers and stores included, measured 6.7 % SLOWER at configs[2] and 8 % at configs[3]: profiles/r05/walk16_paired_tiles_ab.txt.)
template <int NFT_W, int NRT_W, int NT, int D, bool PAIR = false>
__device__ __forceinline__ void mfma_edge_p(k2_gbytes sb, const unsigned (&voff)[NRT_W], unsigned kstride_bytes,
                                            const double* ap, int astride, int ksteps, cafe_d4 (&acc)[NFT_W][NRT_W])
{
    static_assert(!PAIR || NT % 2 == 0, "paired loads carry two row tiles each");
    constexpr int NA_ = NFT_W;
    constexpr bool DS_FIRST_ = true;
    constexpr bool PAIR_ = PAIR;
    k2_lptr pa[NFT_W];
#pragma unroll
    for (int i = 0; i < NFT_W; ++i) pa[i] = (k2_lptr)ap + i * astride;
#define CAFE_K2_MFMA(slot)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NFT_W; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                         \
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[slot][i], bq[slot][j], acc[i][j], 0, 0, 0);
    CAFE_K2_EDGE_BODY(NFT_W, 16)
#undef CAFE_K2_MFMA
}

template <int G, int NRT_W, int NT, int D>
__device__ __forceinline__ void mfma4_edge_p(k2_gbytes sb, const unsigned (&voff)[NRT_W], unsigned kstride_bytes,
                                             const double* ap4, int LDv, int ksteps, double (&acc)[G][NRT_W])
{
    constexpr int NA_ = G;
    constexpr bool DS_FIRST_ = false;
    constexpr bool PAIR_ = false;
    k2_lptr pa[G];
#pragma unroll
    for (int g = 0; g < G; ++g) pa[g] = (k2_lptr)ap4 + (4 * g) * LDv;
#define CAFE_K2_MFMA(slot)                                                                                     \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                                                              \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                         \
            acc[g][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[slot][g], bq[slot][j], acc[g][j], 0, 0, 0);
    CAFE_K2_EDGE_BODY(G, 4)
#undef CAFE_K2_MFMA
}
#undef CAFE_K2_EDGE_BODY
#undef CAFE_K2_LOAD
#undef CAFE_K2_REGION_L
#undef CAFE_K2_REGION_N

// Batch mode trims a tile's row tiles (the walk's prologue), so a wave can be dealt anything from 1 to NRT_W of them:
// one instantiation of the product per live count -- a wave must not issue matrix instructions for columns it does
// not own (round 3's first trimmed launch issued NRT_W columns whatever the count: trimming saved k-steps only).
// (Deeper operand rings for the short regions of these products -- as deep as the registers of the NRT_W-tile product
// allow -- changed nothing for the Monte-Carlo-null launch and cost the objective walk at configs[2] 2.5 %: same box,
// profiles/r03/mcnull_trimmed_counts_grids_mixing.txt.)
template <int NFT_W, int NRT_W, int NT>
__device__ __forceinline__ void mfma_edge_few(int ntile, k2_gbytes sb, const unsigned (&voff)[NRT_W], unsigned kstride_bytes,
                                              const double* ap, int astride, int ksteps, cafe_d4 (&acc)[NFT_W][NRT_W])
{
    if constexpr (NT >= 1) {
        if (ntile == NT) mfma_edge_p<NFT_W, NRT_W, NT, CAFE_K2_DEPTH16>(sb, voff, kstride_bytes, ap, astride, ksteps, acc);
        else mfma_edge_few<NFT_W, NRT_W, NT - 1>(ntile, sb, voff, kstride_bytes, ap, astride, ksteps, acc);
    }
}
template <int G, int NRT_W, int NT>
__device__ __forceinline__ void mfma4_edge_few(int ntile, k2_gbytes sb, const unsigned (&voff)[NRT_W], unsigned kstride_bytes,
                                               const double* ap4, int LDv, int ksteps, double (&acc)[G][NRT_W])
{
    if constexpr (NT >= 1) {
        if (ntile == NT) mfma4_edge_p<G, NRT_W, NT, CAFE_K2_DEPTH4>(sb, voff, kstride_bytes, ap4, LDv, ksteps, acc);
        else mfma4_edge_few<G, NRT_W, NT - 1>(ntile, sb, voff, kstride_bytes, ap4, LDv, ksteps, acc);
    }
}


// ---------------------------------------------------------------------------------------------------
// Pieces shared by the two kernels
// ---------------------------------------------------------------------------------------------------

// Park scratch in global memory (node vectors waiting for their sibling, when they do not fit LDS): a workgroup owns
// one SLOT of it while it runs.  Slots are claimed with one atomic compare-and-swap on a flag array, scanning from
// blockIdx.x % slots (workgroups are dispatched in order, so the first try almost always succeeds) and released at
// the end of the walk.  With ~2x as many slots as the chip holds workgroups the scratch stays small -- 84 MB at the
// configs[2] shape instead of 413 MB with one region per family tile -- and lives in the 256 MB Infinity Cache.  No
// data passes between owners, so exclusivity is all that is needed; a workgroup that finds every slot taken keeps
// scanning while the owners (resident by construction) finish.  Thread 0 publishes the slot in LDS word `s_slot`
// ahead of the prologue's barrier.  Returns -1 when the schedule parks nothing in global memory.
__device__ __forceinline__ int k2_acquire_park_slot(const K2MfmaArgs& a, int* s_slot, int tid)
{
    if (a.n_park_slots <= 0) return -1;
    if (tid == 0) {
        int s = (int)((blockIdx.y * gridDim.x + blockIdx.x) % (unsigned)a.n_park_slots);
        while (atomicCAS(&a.park_flags[s], 0, 1) != 0) {
            s = (s + 1 == a.n_park_slots) ? 0 : s + 1;
            __builtin_amdgcn_s_sleep(1);
        }
        *s_slot = s;
    }
    return 0;
}

__device__ __forceinline__ void k2_release_park_slot(const K2MfmaArgs& a, const int* s_slot, int tid)
{
    // the root step ended with a workgroup barrier behind every read of the parks
    if (a.n_park_slots > 0 && tid == 0) atomicExch(&a.park_flags[*s_slot], 0);
}


// Batch mode, lock-step generations.  A batch launch is tens of resident-chip-fulls of workgroups, each streaming every
// edge matrix of the tree (tens of MB in all) through a 4 MB L2: left alone, the resident workgroups drift apart until
// every one is on a different edge and nothing is reused (85 GB of fabric traffic for 70 MB of inputs, round 2).
// Workgroup b therefore starts only when the generations before its own -- blocks of `gen_size` = as many workgroups
// as the chip holds -- have (all but gen_slack) finished: a generation starts together, its tiles are neighbours in
// the row order (similar extents: similar pace), so the workgroups of an XCD are on the same one or two edges at any
// time and the hot matrix stays in L2.  Only a pace-maker: nothing is communicated, and a workgroup that has waited
// 2 ms goes ahead anyway (dispatch order is not guaranteed).
__device__ __forceinline__ void k2_wait_for_generation(const K2MfmaArgs& a, int tid)
{
    if (!a.gen_done) return;
    if (tid == 0) {
        const int need = (int)(blockIdx.x / (unsigned)a.gen_size) * a.gen_size - a.gen_slack;
        if (need > 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(a.gen_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > 200000) break;   // 2 ms of the 100 MHz counter
            }
        }
    }
    __syncthreads();
}

// Root vectors (in Lbuf) -> per-family posterior (cafe/lambda.cpp:657-689) or packed root rows (batch mode).
// max_posterior = max_i exp(log L_i + log prior_i) as the reference writes it (:681).  Evaluating log and exp for
// every root size is most of the epilogue's time, and all but one or two of those values lose the max by orders of
// magnitude: a first pass finds the largest PRODUCT L_i * prior_i (one multiplication each), and only the root sizes
// whose product is within 1e-9 relative of it -- the log/exp form agrees with the product to ~1e-13 -- get the exact
// exp(log + log) treatment; the maximum over those IS the maximum over all.  The candidates of all the families of a
// wave are collected in an LDS list and evaluated together, one per lane.  Families whose largest product is below
// 1e-290 (underflow would blur the filter) take the plain loop.
// Wave-wide reductions on the vector ALU's data-parallel primitives (xor 1, xor 2, half-row mirror, row mirror, then
// one readlane per 16-lane row): ~25 instructions and no LDS traffic, where a __shfl_xor butterfly is 6 dependent
// ds_bpermute round trips per 32-bit word -- the root scan of one family took 3.4 k cycles with those, 2.2 k with
// these (cfg 2, s_memtime stamps).  Every lane must be active; every lane receives the result.
// (The scan stays ONE family at a time: unrolled over 4 families it ran 1.6x slower -- code executed once per
// workgroup is bound by instruction fetch, not by its dependency chains.)
__device__ __forceinline__ double k2_wave_max(double v)
{
#define CAFE_DPP_MAX(ctrl)                                                                                     \
    {                                                                                                          \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);              \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);              \
        v = fmax(v, __hiloint2double(hi, lo));                                                                 \
    }
    CAFE_DPP_MAX(0xB1) CAFE_DPP_MAX(0x4E) CAFE_DPP_MAX(0x141) CAFE_DPP_MAX(0x140)
#undef CAFE_DPP_MAX
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r[i] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * i + 15), __builtin_amdgcn_readlane(lo, 16 * i + 15));
    return fmax(fmax(r[0], r[1]), fmax(r[2], r[3]));
}

__device__ __forceinline__ int k2_wave_min(int v)
{
#define CAFE_DPP_MIN(ctrl) v = min(v, __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false));
    CAFE_DPP_MIN(0xB1) CAFE_DPP_MIN(0x4E) CAFE_DPP_MIN(0x141) CAFE_DPP_MIN(0x140)
#undef CAFE_DPP_MIN
    return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

// REGS: the root range fits PR registers per lane (R <= 64 * PR), the lane's prior values and root-vector entries
// live in registers for the whole epilogue; PR = 2 (R <= 128, e.g. the 125 root sizes of a table whose largest
// family has 100 members) halves the scan of PR = 4 (R <= 256)
template <bool REGS, int PR>
__device__ __forceinline__ void k2_epilogue_impl(const K2MfmaArgs& a, const double* Lbuf, void* scratch, int fam0,
                                                 size_t out_off, int wave, int lane, int nwaves)
{
    double* const max_lik = a.max_lik + out_off;       // this parameter set's block of the outputs
    double* const max_post = a.max_post + out_off;
    int32_t* const argmax = a.argmax + out_off;
    unsigned* cand = reinterpret_cast<unsigned*>(scratch) + wave * 64;                        // (family << 16) | root index
    unsigned long long* fmaxbits = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned*>(scratch) + 8 * 64);   // [NF]
    const double* prior = a.prior;
    const double* logprior = a.logprior;
    double pr[PR], lpr[PR];
    if (REGS) {
        // one global round trip per wave instead of one per family and pass (L2 latency under load is ~1-2 k cycles)
#pragma unroll
        for (int q = 0; q < PR; ++q) {
            const int i = lane + 64 * q;
            pr[q] = (i < a.R) ? prior[i] : 0.0;
            lpr[q] = (i < a.R) ? logprior[i] : 0.0;
        }
    }
    const int nq = REGS ? PR : (a.R + 63) / 64;
#ifdef CAFE_K2_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K2_STAMP(2 + 6 * a.n_ops + 1);
#endif
    int n_cand = 0;
    unsigned long long fastmask = 0;   // bit t: the wave's t-th family took the filtered path (wave-uniform)

    auto flush = [&]() {
        __threadfence_block();
        if (lane < n_cand) {
            const unsigned c = cand[lane];
            const int f = (int)(c >> 16), i = (int)(c & 0xFFFFu);
            const double p = exp(log(Lbuf[(size_t)f * a.LDv + i]) + logprior[i]);
            atomicMax(&fmaxbits[f], (unsigned long long)__double_as_longlong(p));   // p >= 0: the bit pattern orders like the value
        }
        __threadfence_block();
        n_cand = 0;
    };

    int t = 0;
    for (int f = wave; f < a.NF; f += nwaves, ++t) {
        const int u = fam0 + f;
        if (u >= a.Fu) continue;
        const double* L = Lbuf + (size_t)f * a.LDv;
        double best = -INFINITY, qmax = 0.0;
        int bi = INT_MAX;  // INT_MAX = this lane has seen no element yet
        double vq[PR];     // REGS: the lane's root-vector entries, read once
#pragma unroll
        for (int q = 0; q < (REGS ? PR : 1); ++q) vq[q] = 0.0;
        if (REGS) {
#pragma unroll
            for (int q = 0; q < PR; ++q) {
                const int i = lane + 64 * q;
                if (i < a.R) {
                    const double v = L[i];
                    vq[q] = v;
                    if (bi == INT_MAX || v > best) {
                        best = v;
                        bi = i;
                    }
                    qmax = fmax(qmax, v * pr[q]);
                }
            }
        } else {
            for (int i = lane; i < a.R; i += 64) {
                const double v = L[i];
                if (bi == INT_MAX || v > best) {
                    best = v;
                    bi = i;
                }
                qmax = fmax(qmax, v * prior[i]);
            }
        }
        // first maximum wins (libcommon/mathfunc.c:9-24): the largest value, then the lowest index holding it (a lane
        // without elements carries -inf / INT_MAX and never wins)
        {
            const double lane_best = best;
            best = k2_wave_max(lane_best);
            bi = k2_wave_min(lane_best == best ? bi : INT_MAX);
            qmax = k2_wave_max(qmax);
        }
        if (lane == 0) {
            max_lik[u] = best;
            argmax[u] = bi;
        }
        if (!(qmax >= 1e-290) || t >= 64) {
            // plain form: every root size through log and exp
            double bestp = -INFINITY;
            for (int i = lane; i < a.R; i += 64) bestp = fmax(bestp, exp(log(L[i]) + logprior[i]));
            bestp = k2_wave_max(bestp);
            if (lane == 0) max_post[u] = bestp;
            continue;
        }
        fastmask |= 1ull << t;
        if (lane == 0) fmaxbits[f] = 0ull;
        const double thr = qmax * (1.0 - 1e-9);
#pragma unroll
        for (int q = 0; q < (REGS ? PR : 1); ++q) {
            // (generic form: q enumerates nothing, the while loop below walks the runs)
            if (REGS) {
                const int i = lane + 64 * q;
                const bool is_c = i < a.R && vq[q] * pr[q] >= thr;
                const unsigned long long m = __ballot(is_c);
                if (m != 0) {
                    const int cnt = __popcll(m);
                    if (n_cand + cnt > 64) flush();
                    if (is_c) cand[n_cand + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned)f << 16) | (unsigned)i;
                    n_cand += cnt;
                }
            }
        }
        if (!REGS) {
            for (int i0 = 0; i0 < a.R; i0 += 64) {
                const int i = i0 + lane;
                const bool is_c = i < a.R && L[i] * prior[i] >= thr;
                const unsigned long long m = __ballot(is_c);
                if (m == 0) continue;
                const int cnt = __popcll(m);
                if (n_cand + cnt > 64) flush();
                if (is_c) cand[n_cand + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned)f << 16) | (unsigned)i;
                n_cand += cnt;
            }
        }
    }
    (void)nq;
    K2_STAMP(2 + 6 * a.n_ops + 2);
    if (n_cand) flush();
    K2_STAMP(2 + 6 * a.n_ops + 3);
    // lane t writes the maximum of the wave's t-th family
    {
        const int f = wave + lane * nwaves;
        if (f < a.NF && fam0 + f < a.Fu && ((fastmask >> lane) & 1ull))
            max_post[fam0 + f] = __longlong_as_double((long long)fmaxbits[f]);
    }
}

// The posterior with a LANE per family, for tables of at most 64 root sizes (the reference's test1 table: 30; its example: 42).
// The wave-per-family form above runs a wave's families one after the other, each a chain of ~500 dependent instructions that
// does not shrink with R: 7.1 us of the 36.6 us test1 walk (option k2_skip_epilogue).  Here lane l of EVERY wave owns family
// l (+ 64 per pass) and wave w scans the root sizes i == w (mod waves): the families' chains run side by side.  Partial results
// meet in LDS through integer atomics on the doubles' bit patterns (likelihoods and products are >= +0: the patterns order like
// the values), three barriers in all:
//   pass 1: max_i L_i and max_i L_i * prior_i                                           -> atomicMax
//   pass 2: the lowest i with L_i == max (first maximum wins, libcommon/mathfunc.c:9-24) -> atomicMin; the maximum over the
//           candidates {i : L_i * prior_i >= (1 - 1e-9) max product} of exp(log L_i + log prior_i) (cafe/lambda.cpp:681 as
//           written) -> atomicMax; a family whose largest product is below 1e-290 takes every i.
// The same values as the form above -- the maximum over the candidates IS the maximum over all root sizes, every term is the
// same expression on the same operands (tests: test_small_root_range_epilogue_equals_the_wave_form).  The prior and its
// logarithm of root size `lane` sit in a register of every wave and are read with v_readlane (i is wave-uniform).
// A separate instantiation of the 4-family walk (k2_prune_mfma4<G, NRT_W, true>, k2_walk4s.hip), NOT a branch inside the other
// kernels: compiled in beside the wave form it cost the configs[2] walk 8 % by its code size alone (round 6), and with more than
// 64 root sizes it is slower than the wave form (configs[1] 56.9 -> 58.2 us).
template <int PR>   // R <= 64 * PR
__device__ __forceinline__ void k2_epilogue_small_r(const K2MfmaArgs& a, const double* Lbuf, void* scratch, int fam0, size_t out_off,
                                                    int wave_in, int lane, int nwaves)
{
    unsigned long long* maxbits = reinterpret_cast<unsigned long long*>(scratch);   // [NF] max_i L_i
    unsigned long long* qbits = maxbits + a.NF;                                      // [NF] max_i L_i * prior_i
    unsigned long long* pbits = qbits + a.NF;                                        // [NF] max posterior
    int* amin = reinterpret_cast<int*>(pbits + a.NF);                                // [NF] argmax (lowest index)
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const int tid = wave * 64 + lane, nthreads = nwaves * 64;
    const int R = a.R;   // <= 64 * PR
    double pr[PR], lpr[PR];
#pragma unroll
    for (int q = 0; q < PR; ++q) {
        pr[q] = (lane + 64 * q < R) ? a.prior[lane + 64 * q] : 0.0;
        lpr[q] = (lane + 64 * q < R) ? a.logprior[lane + 64 * q] : 0.0;
    }
    auto bcast = [](const double (&x)[PR], int i) {   // element i of the array spread over the lanes (i wave-uniform)
        double r = 0.0;
#pragma unroll
        for (int q = 0; q < PR; ++q) {
            const double xq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[q]), i & 63),
                                               __builtin_amdgcn_readlane(__double2loint(x[q]), i & 63));
            if (PR == 1 || (i >> 6) == q) r = xq;
        }
        return r;
    };
    __syncthreads();   // (the scratch overlays the walk's counts and park-slot word, which thread 0 has just read)
    for (int f = tid; f < a.NF; f += nthreads) {
        maxbits[f] = 0ull;
        qbits[f] = 0ull;
        pbits[f] = 0ull;
        amin[f] = INT_MAX;
    }
    __syncthreads();
    for (int f0 = 0; f0 < a.NF; f0 += 64) {   // (wave-uniform trips; a lane beyond the last family scans that family and stores nothing)
        const int f = f0 + lane;
        const double* L = Lbuf + (size_t)min(f, a.NF - 1) * a.LDv;
        double best = 0.0, q = 0.0;
        for (int i = wave; i < R; i += nwaves) {
            const double v = L[i];
            best = fmax(best, v);
            q = fmax(q, v * bcast(pr, i));
        }
        if (f < a.NF) {
            atomicMax(&maxbits[f], (unsigned long long)__double_as_longlong(best));
            atomicMax(&qbits[f], (unsigned long long)__double_as_longlong(q));
        }
    }
    __syncthreads();
    for (int f0 = 0; f0 < a.NF; f0 += 64) {
        const int f = f0 + lane, fc = min(f, a.NF - 1);
        const double* L = Lbuf + (size_t)fc * a.LDv;
        const double M = __longlong_as_double((long long)maxbits[fc]);
        const double qmax = __longlong_as_double((long long)qbits[fc]);
        const bool filtered = qmax >= 1e-290;
        const double thr = qmax * (1.0 - 1e-9);
        int bi = INT_MAX;
        double bestp = 0.0;
        for (int i = wave; i < R; i += nwaves) {
            const double v = L[i];
            if (v == M) bi = min(bi, i);
            if (!filtered || v * bcast(pr, i) >= thr) bestp = fmax(bestp, exp(log(v) + bcast(lpr, i)));
        }
        if (f < a.NF) {
            if (bi != INT_MAX) atomicMin(&amin[f], bi);
            atomicMax(&pbits[f], (unsigned long long)__double_as_longlong(bestp));
        }
    }
    __syncthreads();
    for (int f = tid; f < a.NF; f += nthreads) {
        const int u = fam0 + f;
        if (u >= a.Fu) continue;
        a.max_lik[out_off + u] = __longlong_as_double((long long)maxbits[f]);
        a.argmax[out_off + u] = amin[f];
        a.max_post[out_off + u] = __longlong_as_double((long long)pbits[f]);
    }
}

__device__ __forceinline__ void k2_epilogue(const K2MfmaArgs& a, const double* Lbuf, void* scratch, int fam0, size_t out_off,
                                            bool batch, int wave, int lane, int nwaves)
{
    if (batch) {
        for (int f = wave; f < a.NF; f += nwaves) {
            const int u = fam0 + f;
            if (u >= a.Fu) continue;
            const double* L = Lbuf + (size_t)f * a.LDv;
            const int lo = a.root_lo[u] - a.root_min, hi = a.root_hi[u] - a.root_min;
            double* o = a.out_root + a.out_off[u];
            for (int i = lo + lane; i <= hi; i += 64) o[i - lo] = L[i];
        }
        return;
    }
    // (round 6: a LANE per family -- every wave scanning a slice of the root sizes for all families, partial maxima meeting in LDS
    // atomics, the exact exp(log + log) candidates queued and evaluated one per lane -- was built three ways, bit-identical, and
    // is SLOWER wherever R > 64: configs[1] walk 56.9 -> 58.2 us, configs[2] 1.350 -> 1.418 ms; only the reference's test1 table
    // (30 root sizes) gained, 36.1 -> 32.3 us, and with both forms compiled in, the code size alone cost the configs[2] walk 8 %.
    // The ablation (option k2_skip_epilogue) bounds what any epilogue can give back: 5.7 us at configs[1], 7.1 us on test1.
    // profiles/r06/lane_per_family_epilogue_ab.txt, lane_per_family_epilogue.patch)
    // (round 3: a variant carrying four families per wave pass, one per 16-lane row, was bit-identical and SLOWER --
    // walk 56.4 -> 57.9 us at configs[1]: the epilogue is bound by instruction fetch, and the variant is more code)
    if (a.R <= 128) k2_epilogue_impl<true, 2>(a, Lbuf, scratch, fam0, out_off, wave, lane, nwaves);
    else if (a.R <= 256) k2_epilogue_impl<true, 4>(a, Lbuf, scratch, fam0, out_off, wave, lane, nwaves);
    else k2_epilogue_impl<false, 1>(a, Lbuf, scratch, fam0, out_off, wave, lane, nwaves);
}

// OBJ (round 6): the instantiation an OBJECTIVE evaluation runs -- no per-row column limits (batch mode: the Monte-Carlo null, the
// report's per-family ranges) and every error-model leaf folded into the matrices -- carries none of the batch mode's trimming,
// banded error sums and root-row copies: 54.7 -> 43.2 KB of code and 226 -> 215 registers for k2_prune_mfma4<5, 3>, and the
// objective walk is 1-6 % faster for it (configs[1] 56.7 -> 55.5 us, test1 32.1 -> 30.2 us, the configs[3] shard 0.990 ->
// 0.980 ms; profiles/r06/objective_only_walk_ab.txt): these kernels feel their code size (the CU pair's instruction cache is
// 64 KB).  Same instructions on the same operands for everything an objective evaluation executes: bit-identical.
template <int NFT_W, int NRT_W, bool OBJ = false>
__global__ __launch_bounds__(512) void k2_prune_mfma(K2MfmaArgs a)
{
    extern __shared__ double Lbuf[];                          // [NF][LDv]
    // the walk's step list and each step's matrix index, copied to LDS once: the step loop then never
    // waits on a chain of dependent global loads (step record -> node_key -> matrix base)
    int* s_ops = reinterpret_cast<int*>(Lbuf + (size_t)a.NF * a.LDv * (1 + a.lds_parks));   // [n_ops][12]  (cafehip::MfmaOp)
    int* s_key = s_ops + a.n_ops * 12;                        // [n_ops][2] matrix offsets
    int* s_err = s_key + a.n_ops * 2;                         // [n_ops][2]  1: leaf child that carries the error model
    int* s_cnt = s_err + a.n_ops * 2;                         // [NF][n_leaves]   (per tile; the epilogue's scratch overlays it)
    int* s_colmax = s_cnt + a.NF * a.n_leaves;                // [NF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;   // row within a tile (B/D) or family within a tile (A)
    const int lk = lane >> 4;   // k within a step (A/B) or family group within a tile (D)
    K2_STAMP(0);
    const int wf = wave % a.Wf;
    // workgroups of the second dispatch round share a CU with one of the first: rotate their row-tile deal by
    // half so that the waves carrying the odd extra tile do not pile up on the same SIMDs
    const int wr = (wave / a.Wf + ((blockIdx.x >> 8) & 1) * (a.Wr >> 1)) % a.Wr;
    const int ft0 = wf * NFT_W;
    const bool batch = OBJ ? false : (a.col_max != nullptr);
    const bool fold = OBJ ? true : ((a.PTfold != nullptr) && !batch);   // (OBJ: the launcher guarantees PTfold wherever a leaf carries the model)
    const size_t park_stride = (size_t)a.NF * a.LDv;
    const int fam0 = blockIdx.x * a.NF;
    k2_wait_for_generation(a, tid);
    const int my_slot = k2_acquire_park_slot(a, s_colmax + a.NF, tid);   // (the word behind the column limits)

    for (int i = tid; i < a.n_ops * 12; i += blockDim.x) s_ops[i] = reinterpret_cast<const int*>(a.ops)[i];
    for (int i = tid; i < a.n_ops * 2; i += blockDim.x) {
        const cafehip::MfmaOp& o = a.ops[i >> 1];
        // element offset (< 2^31) of the child's matrix, or of its factor table when it is a compressed subtree
        s_key[i] = (o.kind[i & 1] == 2) ? a.table_off[o.child[i & 1]] : a.node_key[blockIdx.y * a.n_nodes + o.child[i & 1]] * a.KP * a.LD;
        s_err[i] = (o.kind[i & 1] == 0 && a.err != nullptr && a.leaf_has_err[o.leafcol[i & 1]]) ? 1 : 0;
    }

    for (int i = tid; i < a.NF * a.n_leaves; i += blockDim.x) {
        const int f = i / a.n_leaves, j = i - f * a.n_leaves;
        const int u = fam0 + f;
        s_cnt[i] = (u < a.Fu) ? a.counts[(size_t)u * a.n_leaves + j] : 0;
    }
    for (int f = tid; f < a.NF; f += blockDim.x) {
        const int u = fam0 + f;
        // (batch mode: slots behind the last row carry limit 0, so that they do not widen the tile's trimmed extent)
        s_colmax[f] = batch ? ((u < a.Fu) ? a.col_max[u] : 0) : (a.C - 1);
    }
    __syncthreads();
    // Batch mode (per-row column limits, cafe/conditional_distribution.cpp:16-32): rows beyond a family's limit are
    // zero in every node vector, so a product over k-steps or row tiles beyond the tile's LARGEST limit multiplies and
    // produces zeros only.  The tile's products stop there (same values: the skipped matrix instructions add exact
    // zeros), and its root step covers only the row tiles that hold a root size some family of the tile asks for.
    int t_ksteps = a.ksteps, t_rt = INT_MAX, t_root_first = 0, t_root_last = INT_MAX;
    if (batch && a.trim) {
        int kmax = 0, rlo = INT_MAX, rhi = 0;
        for (int f = 0; f < a.NF; ++f) {
            kmax = max(kmax, s_colmax[f]);
            const int u = fam0 + f;
            if (u < a.Fu) {
                rlo = min(rlo, a.root_lo[u] - a.root_min);
                rhi = max(rhi, a.root_hi[u] - a.root_min);
            }
        }
        t_ksteps = min(a.ksteps, (kmax + 4) >> 2);
        t_rt = (kmax + 16) >> 4;
        t_root_first = min(rlo, rhi) >> 4;
        t_root_last = rhi >> 4;
    }
    K2_STAMP(1);
    double* my_park = a.park + (size_t)(my_slot >= 0 ? s_colmax[a.NF] : (int)(blockIdx.y * gridDim.x + blockIdx.x)) * a.n_parks * park_stride;

    // this lane's families do not change during the walk: their column limits are read once, not once per step
    int cmx[NFT_W][4];
#pragma unroll
    for (int i = 0; i < NFT_W; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmx[i][r] = OBJ ? a.C - 1 : s_colmax[(ft0 + i) * 16 + lk + 4 * r];   // (OBJ: one limit, a scalar)

    for (int oi = 0; oi < a.n_ops; ++oi) {
        const cafehip::MfmaOp op = *reinterpret_cast<const cafehip::MfmaOp*>(s_ops + oi * 12);
        const int rows = op.is_root ? a.R : a.C;
        const int row_lo = op.is_root ? a.root_min : 0;
        // row tiles of this step (trimmed in batch mode), dealt evenly to the Wr wave rows
        const int rt_first = op.is_root ? t_root_first : 0;
        const int RT = op.is_root ? (min(t_root_last, ((rows + 15) >> 4) - 1) + 1 - rt_first) : min((rows + 15) >> 4, t_rt);
        const int rt_base = RT / a.Wr, rt_rem = RT - rt_base * a.Wr;
        const int ntile = rt_base + (wr < rt_rem ? 1 : 0);  // <= NRT_W
        const int rt0 = rt_first + wr * rt_base + min(wr, rt_rem);
        const bool wave_active = ntile > 0;

        cafe_d4 hold[NFT_W][NRT_W];   // declared per step: nothing of it is live across steps
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const bool has_err = s_err[oi * 2 + ch] != 0;
            const bool folded = has_err && fold;   // gathers on the folded matrix, like a one-hot leaf
            const bool errleaf = has_err && !fold;
            const bool from_table = op.kind[ch] == 2;   // compressed subtree: the factor is a row of its table
            const double* PTe = (from_table ? a.tables + (size_t)blockIdx.y * a.table_set_stride : (folded ? a.PTfold : a.PT)) +
                                s_key[oi * 2 + ch] + row_lo;
            cafe_d4 fac[NFT_W][NRT_W];
            if (errleaf && a.err_banded) {
                // banded error model (as read from a model file, cafe/error_model.cpp:162-189): the leaf
                // vector has a handful of non-zeros around the observed count, so the edge product is a
                // short sum of column gathers, k ascending, instead of a GEMM
#pragma unroll
                for (int i = 0; i < NFT_W; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = (ft0 + i) * 16 + lk + 4 * r;
                        const int cnt = s_cnt[f * a.n_leaves + op.leafcol[ch]];
                        const int klo = max(cnt + a.err_dlo, 0);
                        const int khi = min(min(cnt + a.err_dhi, a.C - 1), cmx[i][r]);
                        const double* erow = a.err + (size_t)cnt * a.err_ld;
#pragma unroll
                        for (int j = 0; j < NRT_W; ++j) {
                            double v = 0.0;
                            if (j < ntile)
                                for (int k = klo; k <= khi; ++k)
                                    v += erow[k] * PTe[(size_t)k * a.LD + (rt0 + j) * 16 + li];
                            fac[i][j][r] = v;
                        }
                    }
                }
            } else if (op.kind[ch] != 1 && !errleaf) {
                // one-hot leaf: factor = PT[count][row]  (cafe/cafe_tree.c:208-209); compressed subtree: table[state][row]
#pragma unroll
                for (int i = 0; i < NFT_W; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = (ft0 + i) * 16 + lk + 4 * r;
                        const int cnt = s_cnt[f * a.n_leaves + op.leafcol[ch]];
                        const bool ok = OBJ || from_table || cnt <= cmx[i][r];   // (OBJ: counts never exceed the range)
                        // one address per family; the row tiles are constant byte offsets from it
                        const double* col = PTe + (size_t)cnt * a.LD + rt0 * 16 + li;
#pragma unroll
                        for (int j = 0; j < NRT_W; ++j) fac[i][j][r] = (ok && j < ntile) ? col[j * 16] : 0.0;
                    }
                }
            } else {
                const double* Lsrc = Lbuf;
                if (op.kind[ch] == 1 && op.src_park[ch] >= 0 && op.src_park[ch] < a.lds_parks) {
                    Lsrc = Lbuf + (size_t)(1 + op.src_park[ch]) * park_stride;  // read the parked vector in place
                } else if (op.kind[ch] == 1 && op.src_park[ch] >= 0) {
                    // fetch the parked vector into the LDS buffer
                    __syncthreads();
                    const double* src = my_park + (size_t)op.src_park[ch] * park_stride;
                    if (batch && a.trim) {
                        // a trimmed tile reads only what its products use, entries [0, 4 * t_ksteps) of every row (no faster
                        // -- the fetch is latency -- but 1.4 GB less fabric traffic for the Monte-Carlo null)
                        const int kk = min(4 * t_ksteps, a.LDv);
                        for (int i = tid; i < a.NF * kk; i += blockDim.x) {
                            const int f = i / kk, k = i - f * kk;
                            Lbuf[f * a.LDv + k] = src[f * a.LDv + k];
                        }
                    } else {
                        for (int i = tid; i < a.NF * a.LDv; i += blockDim.x) Lbuf[i] = src[i];
                    }
                    __syncthreads();
                } else if (errleaf) {
                    // leaf vector = errormatrix[observed][0..C)  (cafe/cafe_tree.c:196-203)
                    __syncthreads();
                    for (int i = tid; i < a.NF * a.LDv; i += blockDim.x) {
                        const int f = i / a.LDv, k = i - f * a.LDv;
                        const int cnt = s_cnt[f * a.n_leaves + op.leafcol[ch]];
                        Lbuf[i] = (k < a.C && k <= s_colmax[f]) ? a.err[(size_t)cnt * a.err_ld + k] : 0.0;
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int i = 0; i < NFT_W; ++i)
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) fac[i][j] = cafe_d4{0.0, 0.0, 0.0, 0.0};
                if (wave_active) {
                    // matrix operand: uniform base + this lane's (k row, matrix row) and the wave's row tiles
                    unsigned voff[NRT_W];
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j)   // inactive tiles re-read tile rt0
                        voff[j] = (unsigned)(lk * a.LD + li + ((j < ntile) ? (rt0 + j) : rt0) * 16) * 8u;
                    const k2_gbytes sb = k2_uniform(PTe);
                    const double* ap = Lsrc + (size_t)(ft0 * 16 + li) * a.LDv + lk;
                    const unsigned kstride_bytes = 32u * (unsigned)a.LD;
                    if constexpr (NRT_W > 1) {
                        if (ntile == NRT_W)
                            mfma_edge_p<NFT_W, NRT_W, NRT_W, CAFE_K2_DEPTH16>(sb, voff, kstride_bytes, ap, 16 * a.LDv, t_ksteps, fac);
                        else if (ntile == NRT_W - 1)
                            mfma_edge_p<NFT_W, NRT_W, NRT_W - 1, CAFE_K2_DEPTH16>(sb, voff, kstride_bytes, ap, 16 * a.LDv, t_ksteps, fac);
                        else   // a trimmed tile of a batch launch
                            mfma_edge_few<NFT_W, NRT_W, NRT_W - 2>(ntile, sb, voff, kstride_bytes, ap, 16 * a.LDv, t_ksteps, fac);
                    } else {
                        mfma_edge_p<NFT_W, NRT_W, NRT_W, CAFE_K2_DEPTH16>(sb, voff, kstride_bytes, ap, 16 * a.LDv, t_ksteps, fac);
                    }
                }
            }
            K2_STAMP(2 + 6 * oi + 1 + ch);
            if (ch == 0) {
#pragma unroll
                for (int i = 0; i < NFT_W; ++i)
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) hold[i][j] = fac[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < NFT_W; ++i)
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) hold[i][j] *= fac[i][j];
            }
        }

        // ---- result: Hadamard product in `hold` (D layout) -> LDS buffer or park ----
        // (two explicit address spaces: a pointer that may be either makes every store a flat_store)
        if (op.dst_park >= a.lds_parks) {
            double* dst = my_park + (size_t)op.dst_park * park_stride;
#pragma unroll
            for (int i = 0; i < NFT_W; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = (ft0 + i) * 16 + lk + 4 * r;
                    const int cm = cmx[i][r];
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) {
                        if (j < ntile) {
                            const int row = (rt0 + j) * 16 + li;
                            // rows beyond this family's column range do not exist in the reference
                            dst[(size_t)f * a.LDv + row] = (!op.is_root && row > cm) ? 0.0 : hold[i][j][r];
                        }
                    }
                }
            }
        } else {
            __syncthreads();  // every wave is done reading the buffers: overwrite in place
            K2_STAMP(2 + 6 * oi + 3);
            double* dst = Lbuf + ((op.dst_park >= 0) ? (size_t)(1 + op.dst_park) * park_stride : 0);
            const int row0 = rt0 * 16 + li;
#pragma unroll
            for (int i = 0; i < NFT_W; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = (ft0 + i) * 16 + lk + 4 * r;
                    double* d = dst + f * a.LDv + row0;   // row tiles: constant offsets
                    const int lim = op.is_root ? INT_MAX : cmx[i][r];
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j)
                        if (j < ntile) d[j * 16] = (row0 + j * 16 > lim) ? 0.0 : hold[i][j][r];
                }
            }
            __syncthreads();
        }
        K2_STAMP(2 + 6 * oi + 4);
    }

    // ---- root vector (in Lbuf) -> posterior (cafe/lambda.cpp:657-689) or packed root rows ----
    k2_release_park_slot(a, s_colmax + a.NF, tid);
    if (a.gen_done && tid == 0) atomicAdd(a.gen_done, 1);
    if (a.skip_epilogue) return;   // (ablation only: what the posterior epilogue costs the launch)
    k2_epilogue(a, Lbuf, s_cnt, fam0, (size_t)blockIdx.y * a.Fu, batch, wave, lane, blockDim.x >> 6);
    K2_STAMP(2 + 6 * a.n_ops);
}


// ====================================================================================
// K2 with 4-family granularity: v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction,
// 16 cycles, the same 32 flop/cycle/SIMD as the 64-cycle 16x16x4; measured SQ_VALU_MFMA_BUSY_CYCLES per
// instruction: 16.0 and 64.0).  Mapping the 4 blocks to 4 groups of 4 rows makes the B operand IDENTICAL to
// the 16x16x4 one (lane l: PT[k0 + (l>>4)][row0 + (l&15)]); with the A operand carrying families
// fam + (l&3) (lane l: L[fam + (l&3)][k0 + (l>>4)], broadcast over the blocks) the result lane l is
// Y[fam + (l>>4)][row0 + (l&15)] -- one register of the 16x16 accumulator tile (layout decoded by
// tools/mfma_f64_4x4_probe.hip).  A workgroup owns NF = 4*G*Wf
// families (any multiple of 4, not only of 16), so the family tiles can be sized to fill the 256
// CUs evenly when the table is small (10 k families = 625 tiles of 16 put 3 tiles on 113 CUs and 2 on
// the rest; 250 tiles of 40 put one on each of 250 CUs).  Accumulator g of a wave holds, in lane l,
// Y[fam_base + 4g + (l>>4)][row0 + (l&15)]: the same walk, gathers and stores as k2_prune_mfma with
// (i, r) flattened to g = 4i + r.
// ====================================================================================
template <int G, int NRT_W, int SMALL_R = 0, bool OBJ = (SMALL_R > 0)>   // SMALL_R > 0: lane-per-family epilogue for R <= 64 * SMALL_R; OBJ: see k2_prune_mfma
__global__ __launch_bounds__(512) void k2_prune_mfma4(K2MfmaArgs a)
{
    extern __shared__ double Lbuf[];                          // [NF][LDv]
    // the walk's step list and each step's matrix index, copied to LDS once: the step loop then never
    // waits on a chain of dependent global loads (step record -> node_key -> matrix base)
    int* s_ops = reinterpret_cast<int*>(Lbuf + (size_t)a.NF * a.LDv * (1 + a.lds_parks));   // [n_ops][12]  (cafehip::MfmaOp)
    int* s_key = s_ops + a.n_ops * 12;                        // [n_ops][2] matrix offsets
    int* s_err = s_key + a.n_ops * 2;                         // [n_ops][2]  1: leaf child that carries the error model
    int* s_cnt = s_err + a.n_ops * 2;                         // [NF][n_leaves]   (per tile; the epilogue's scratch overlays it)
    int* s_colmax = s_cnt + a.NF * a.n_leaves;                // [NF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lk = lane >> 4;
    K2_STAMP(0);
    const int wf = wave % a.Wf;
    // workgroups of the second dispatch round share a CU with one of the first: rotate their row-tile deal by
    // half so that the waves carrying the odd extra tile do not pile up on the same SIMDs
    const int wr = (wave / a.Wf + ((blockIdx.x >> 8) & 1) * (a.Wr >> 1)) % a.Wr;
    const int fbase = wf * 4 * G;               // first family of this wave inside the workgroup
    const bool batch = OBJ ? false : (a.col_max != nullptr);
    const bool fold = OBJ ? true : ((a.PTfold != nullptr) && !batch);   // (OBJ: the launcher guarantees PTfold wherever a leaf carries the model)
    const size_t park_stride = (size_t)a.NF * a.LDv;
    const int fam0 = blockIdx.x * a.NF;
    k2_wait_for_generation(a, tid);
    const int my_slot = k2_acquire_park_slot(a, s_colmax + a.NF, tid);   // (the word behind the column limits)

    for (int i = tid; i < a.n_ops * 12; i += blockDim.x) s_ops[i] = reinterpret_cast<const int*>(a.ops)[i];
    for (int i = tid; i < a.n_ops * 2; i += blockDim.x) {
        const cafehip::MfmaOp& o = a.ops[i >> 1];
        // element offset (< 2^31) of the child's matrix, or of its factor table when it is a compressed subtree
        s_key[i] = (o.kind[i & 1] == 2) ? a.table_off[o.child[i & 1]] : a.node_key[blockIdx.y * a.n_nodes + o.child[i & 1]] * a.KP * a.LD;
        s_err[i] = (o.kind[i & 1] == 0 && a.err != nullptr && a.leaf_has_err[o.leafcol[i & 1]]) ? 1 : 0;
    }

    for (int i = tid; i < a.NF * a.n_leaves; i += blockDim.x) {
        const int f = i / a.n_leaves, j = i - f * a.n_leaves;
        const int u = fam0 + f;
        s_cnt[i] = (u < a.Fu) ? a.counts[(size_t)u * a.n_leaves + j] : 0;
    }
    for (int f = tid; f < a.NF; f += blockDim.x) {
        const int u = fam0 + f;
        // (batch mode: slots behind the last row carry limit 0, so that they do not widen the tile's trimmed extent)
        s_colmax[f] = batch ? ((u < a.Fu) ? a.col_max[u] : 0) : (a.C - 1);
    }
    __syncthreads();
    // Batch mode (per-row column limits, cafe/conditional_distribution.cpp:16-32): rows beyond a family's limit are
    // zero in every node vector, so a product over k-steps or row tiles beyond the tile's LARGEST limit multiplies and
    // produces zeros only.  The tile's products stop there (same values: the skipped matrix instructions add exact
    // zeros), and its root step covers only the row tiles that hold a root size some family of the tile asks for.
    int t_ksteps = a.ksteps, t_rt = INT_MAX, t_root_first = 0, t_root_last = INT_MAX;
    if (batch && a.trim) {
        int kmax = 0, rlo = INT_MAX, rhi = 0;
        for (int f = 0; f < a.NF; ++f) {
            kmax = max(kmax, s_colmax[f]);
            const int u = fam0 + f;
            if (u < a.Fu) {
                rlo = min(rlo, a.root_lo[u] - a.root_min);
                rhi = max(rhi, a.root_hi[u] - a.root_min);
            }
        }
        t_ksteps = min(a.ksteps, (kmax + 4) >> 2);
        t_rt = (kmax + 16) >> 4;
        t_root_first = min(rlo, rhi) >> 4;
        t_root_last = rhi >> 4;
    }
    K2_STAMP(1);
    double* my_park = a.park + (size_t)(my_slot >= 0 ? s_colmax[a.NF] : (int)(blockIdx.y * gridDim.x + blockIdx.x)) * a.n_parks * park_stride;

    // this lane's families do not change during the walk: their column limits are read once, not once per step
    int cmx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) cmx[g] = OBJ ? a.C - 1 : s_colmax[fbase + 4 * g + lk];   // (OBJ: one limit, a scalar)

    for (int oi = 0; oi < a.n_ops; ++oi) {
        const cafehip::MfmaOp op = *reinterpret_cast<const cafehip::MfmaOp*>(s_ops + oi * 12);
        const int rows = op.is_root ? a.R : a.C;
        const int row_lo = op.is_root ? a.root_min : 0;
        const int rt_first = op.is_root ? t_root_first : 0;   // (batch mode: trimmed extents, see the prologue)
        const int RT = op.is_root ? (min(t_root_last, ((rows + 15) >> 4) - 1) + 1 - rt_first) : min((rows + 15) >> 4, t_rt);
        const int rt_base = RT / a.Wr, rt_rem = RT - rt_base * a.Wr;
        const int ntile = rt_base + (wr < rt_rem ? 1 : 0);
        const int rt0 = rt_first + wr * rt_base + min(wr, rt_rem);
        const bool wave_active = ntile > 0;

        // one-hot leaf child next to a child that needs the matrix cores: issue its column gathers first so
        // that their latency hides under the sibling's product (the Hadamard product commutes exactly)
        bool simple[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
            simple[ch] = (op.kind[ch] != 1) && (fold || !s_err[oi * 2 + ch]);
        const int pre_ch = (simple[1] && !simple[0]) ? 1 : ((simple[0] && !simple[1]) ? 0 : -1);
        double pre[G][NRT_W];
        if (pre_ch >= 0) {
            const int leafcol = pre_ch ? op.leafcol[1] : op.leafcol[0];
            const bool pre_folded = fold && s_err[oi * 2 + pre_ch] != 0;
            const bool pre_table = (pre_ch ? op.kind[1] : op.kind[0]) == 2;
            const double* PTe = (pre_table ? a.tables + (size_t)blockIdx.y * a.table_set_stride : (pre_folded ? a.PTfold : a.PT)) +
                                s_key[oi * 2 + pre_ch] + row_lo;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int f = fbase + 4 * g + lk;
                const int cnt = s_cnt[f * a.n_leaves + leafcol];
                const bool ok = OBJ || pre_table || cnt <= cmx[g];   // (OBJ: counts never exceed the range)
                // one address per family group; the row tiles are constant byte offsets from it
                const double* col = PTe + (size_t)cnt * a.LD + rt0 * 16 + li;
#pragma unroll
                for (int j = 0; j < NRT_W; ++j) pre[g][j] = (ok && j < ntile) ? col[j * 16] : 0.0;
            }
        }

        K2_STAMP(2 + 6 * oi + 0);
        double hold[G][NRT_W];   // declared per step: nothing of it is live across steps
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < NRT_W; ++j) hold[g][j] = 1.0;
        bool first = true;
#pragma unroll   // (rolled -- one copy of the product code for both children, 43 -> 36 KB -- the kernel spills and the configs[1] walk goes 55 -> 63 us)
        for (int ch = 0; ch < 2; ++ch) {
            if (ch == pre_ch) continue;
            const bool has_err = s_err[oi * 2 + ch] != 0;
            const bool folded = has_err && fold;
            const bool errleaf = has_err && !fold;
            const bool from_table = op.kind[ch] == 2;
            const double* PTe = (from_table ? a.tables + (size_t)blockIdx.y * a.table_set_stride : (folded ? a.PTfold : a.PT)) +
                                s_key[oi * 2 + ch] + row_lo;
            double fac[G][NRT_W];
            if (errleaf && a.err_banded) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int f = fbase + 4 * g + lk;
                    const int cnt = s_cnt[f * a.n_leaves + op.leafcol[ch]];
                    const int klo = max(cnt + a.err_dlo, 0);
                    const int khi = min(min(cnt + a.err_dhi, a.C - 1), cmx[g]);
                    const double* erow = a.err + (size_t)cnt * a.err_ld;
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) {
                        double v = 0.0;
                        if (j < ntile)
                            for (int k = klo; k <= khi; ++k) v += erow[k] * PTe[(size_t)k * a.LD + (rt0 + j) * 16 + li];
                        fac[g][j] = v;
                    }
                }
            } else if (op.kind[ch] != 1 && !errleaf) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int f = fbase + 4 * g + lk;
                    const int cnt = s_cnt[f * a.n_leaves + op.leafcol[ch]];
                    const bool ok = OBJ || from_table || cnt <= cmx[g];
                    const double* col = PTe + (size_t)cnt * a.LD + rt0 * 16 + li;
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) fac[g][j] = (ok && j < ntile) ? col[j * 16] : 0.0;
                }
            } else {
                const double* Lsrc = Lbuf;
                if (op.kind[ch] == 1 && op.src_park[ch] >= 0 && op.src_park[ch] < a.lds_parks) {
                    Lsrc = Lbuf + (size_t)(1 + op.src_park[ch]) * park_stride;
                } else if (op.kind[ch] == 1 && op.src_park[ch] >= 0) {
                    __syncthreads();
                    const double* src = my_park + (size_t)op.src_park[ch] * park_stride;
                    if (batch && a.trim) {
                        // a trimmed tile reads only what its products use, entries [0, 4 * t_ksteps) of every row (no faster
                        // -- the fetch is latency -- but 1.4 GB less fabric traffic for the Monte-Carlo null)
                        const int kk = min(4 * t_ksteps, a.LDv);
                        for (int i = tid; i < a.NF * kk; i += blockDim.x) {
                            const int f = i / kk, k = i - f * kk;
                            Lbuf[f * a.LDv + k] = src[f * a.LDv + k];
                        }
                    } else {
                        for (int i = tid; i < a.NF * a.LDv; i += blockDim.x) Lbuf[i] = src[i];
                    }
                    __syncthreads();
                } else if (errleaf) {
                    __syncthreads();
                    for (int i = tid; i < a.NF * a.LDv; i += blockDim.x) {
                        const int f = i / a.LDv, k = i - f * a.LDv;
                        const int cnt = s_cnt[f * a.n_leaves + op.leafcol[ch]];
                        Lbuf[i] = (k < a.C && k <= s_colmax[f]) ? a.err[(size_t)cnt * a.err_ld + k] : 0.0;
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j) fac[g][j] = 0.0;
                if (wave_active) {
                    unsigned voff[NRT_W];
#pragma unroll
                    for (int j = 0; j < NRT_W; ++j)
                        voff[j] = (unsigned)(lk * a.LD + li + ((j < ntile) ? (rt0 + j) : rt0) * 16) * 8u;
                    const k2_gbytes sb = k2_uniform(PTe);
                    const double* ap4 = Lsrc + (size_t)(fbase + (lane & 3)) * a.LDv + lk;
                    const unsigned kstride_bytes = 32u * (unsigned)a.LD;
                    if constexpr (NRT_W > 1) {
                        if (ntile == NRT_W)
                            mfma4_edge_p<G, NRT_W, NRT_W, CAFE_K2_DEPTH4>(sb, voff, kstride_bytes, ap4, a.LDv, t_ksteps, fac);
                        else if (ntile == NRT_W - 1)
                            mfma4_edge_p<G, NRT_W, NRT_W - 1, CAFE_K2_DEPTH4>(sb, voff, kstride_bytes, ap4, a.LDv, t_ksteps, fac);
                        else   // a trimmed tile of a batch launch
                            mfma4_edge_few<G, NRT_W, NRT_W - 2>(ntile, sb, voff, kstride_bytes, ap4, a.LDv, t_ksteps, fac);
                    } else {
                        mfma4_edge_p<G, NRT_W, NRT_W, CAFE_K2_DEPTH4>(sb, voff, kstride_bytes, ap4, a.LDv, t_ksteps, fac);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NRT_W; ++j) hold[g][j] = first ? fac[g][j] : hold[g][j] * fac[g][j];   // (1.0 * x == x exactly)
            first = false;
            K2_STAMP(2 + 6 * oi + 1 + ch);
        }
        if (pre_ch >= 0) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NRT_W; ++j) hold[g][j] *= pre[g][j];
        }

        if (op.dst_park >= a.lds_parks) {
            double* dst = my_park + (size_t)op.dst_park * park_stride;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int f = fbase + 4 * g + lk;
                const int cm = cmx[g];
#pragma unroll
                for (int j = 0; j < NRT_W; ++j) {
                    if (j < ntile) {
                        const int row = (rt0 + j) * 16 + li;
                        dst[(size_t)f * a.LDv + row] = (!op.is_root && row > cm) ? 0.0 : hold[g][j];
                    }
                }
            }
        } else {
            __syncthreads();
            K2_STAMP(2 + 6 * oi + 3);
            double* dst = Lbuf + ((op.dst_park >= 0) ? (size_t)(1 + op.dst_park) * park_stride : 0);
            const int row0 = rt0 * 16 + li;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                double* d = dst + (fbase + 4 * g + lk) * a.LDv + row0;   // row tiles: constant offsets
                const int lim = op.is_root ? INT_MAX : cmx[g];
#pragma unroll
                for (int j = 0; j < NRT_W; ++j)
                    if (j < ntile) d[j * 16] = (row0 + j * 16 > lim) ? 0.0 : hold[g][j];
            }
            __syncthreads();
        }
        K2_STAMP(2 + 6 * oi + 4);
    }

    k2_release_park_slot(a, s_colmax + a.NF, tid);
    if (a.gen_done && tid == 0) atomicAdd(a.gen_done, 1);
    if (a.skip_epilogue) return;   // (ablation only: what the posterior epilogue costs the launch)
    if constexpr (SMALL_R > 0) k2_epilogue_small_r<SMALL_R>(a, Lbuf, s_cnt, fam0, (size_t)blockIdx.y * a.Fu, wave, lane, blockDim.x >> 6);
    else k2_epilogue(a, Lbuf, s_cnt, fam0, (size_t)blockIdx.y * a.Fu, batch, wave, lane, blockDim.x >> 6);
    K2_STAMP(2 + 6 * a.n_ops);
}


// ====================================================================================
// k2c_nodes -- factor tables of compressed subtrees (schedule.hpp, CTile), one launch per level.
// A workgroup owns 16 states of one node: it forms their node vectors in LDS, L[state][k] = F_a[k] * F_b[k] with
// F_x a matrix column (leaf child: PT[count], folded with the error model where the leaf carries one) or a row of
// the child's table, multiplies them by the node's own edge matrix with the walk's edge product (same k order, same
// instruction: the table rows are bit-identical to the factors the uncompressed walk forms in registers) and stores
// rows [0, C) of the result as table[state][row]; the consumer adds the root offset exactly as for a leaf column.
// 16x16x4 shape, one 16-state tile, Wr = blockDim.x / 64 wave rows of NRT_W row tiles.
// ====================================================================================

// A level is a few hundred workgroups: one or two waves per SIMD, nothing to switch to while an operand is in flight,
// so the rings are deep and a tile's rows are spread over as many waves as it has row tiles (up to 16).  A tile holds
// 16 * NFT_W states; the launcher uses NFT_W = 1 (larger tiles measured slower).  The child columns of a state are
// requested in ONE batch (clamped addresses, no branches) and every index the tile needs from the evaluation's
// node -> matrix map in one round trip behind the tile record.
// Measured and dropped in round 3 (profiles/r03/k2c_prefetch_sweep.txt): requesting the wave's whole matrix operand
// up front (8 ... 40 k-steps in registers, straight-line product): 19-47 % SLOWER tables at every bench shape -- a
// level is bound by how many tiles a CU holds at once (registers) and by the matrix pipe, not by operand latency.
// Ring depth (sweep of 4 / 6 / 8 / 12 / 16 in round 3, profiles/r03/k1_k3_k2c_micro_sweeps.txt): 4 and 6 tie at configs[1] and
// [3], 4 is 3 % faster at configs[2]; 12 and 16 are 6-20 % slower (registers, not latency, bound a level).
#ifndef CAFE_K2C_DEPTH
#define CAFE_K2C_DEPTH 4
#endif
constexpr int K2C_GATHER_MAX = 6;   // vector slices per thread kept in registers (LDv <= 6 * threads per state, else a loop)

template <int NFT_W, int NRT_W, bool BATCH, bool PAIR = false>
__global__ __launch_bounds__(1024) void k2c_nodes(K2cArgs a)
{
    extern __shared__ double Lbuf[];   // [16 * NFT_W][LDv]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    // A tile's set-up is a chain of dependent loads (tile record -> node -> matrix map -> columns) at ~1 us each (the
    // tables written by the previous level have flushed L2: every stage is an Infinity-Cache round trip), and a level of
    // a small table lasts as long as ONE tile.  The chain is two round trips before the product:
    //   1. (needs blockIdx only) the tile's header, this thread's two child indices, AND the whole node -> matrix map
    //      and error-model flags of a tree of up to 128 nodes, one entry per lane -- the three entries the tile needs
    //      are then picked out of registers (v_readlane) instead of a second dependent round trip;
    //      (larger trees: the five lookups as a second stage, all unconditionally, clamped where a field does not apply)
    //   2. the gathers.
    // (Left to itself the compiler sinks each lookup into the branch that uses it: eight serialised waits, round 3.)
    // every argument requested at once (the compiler otherwise fetches the kernel arguments in two dependent batches)
    asm volatile("" ::"s"(a.PT), "s"(a.PTfold), "s"(a.node_key), "s"(a.n_nodes), "s"(a.tiles), "s"(a.leaf_has_err32), "s"(a.tables),
                 "s"(a.table_set_stride), "s"(a.C), "s"(a.LD), "s"(a.KP), "s"(a.LDv), "s"(a.ksteps), "s"(a.block_threads));
    K2C_STAMP(0);
    const cafehip::CTile& t = a.tiles[blockIdx.x];
    const int set = blockIdx.y;
    const int per_state = a.block_threads / (16 * NFT_W);   // blockDim / states threads per state (all threads busy; contiguous runs of both columns)
    const int f = tid / per_state, l = tid - f * per_state;   // f < 16 * NFT_W
    const int i0 = t.idx[0][f], i1 = t.idx[1][f];
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
    const int ntile = rt_base + (wave < rt_rem ? 1 : 0);
    const int rt0 = wave * rt_base + min(wave, rt_rem);
    const unsigned kstride_bytes = 32u * (unsigned)a.LD;
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
    {
        // L[state][k] = F_a[k] * F_b[k]
        const bool fold0 = a.PTfold != nullptr && a.leaf_has_err32 != nullptr && err0;
        const bool fold1 = a.PTfold != nullptr && a.leaf_has_err32 != nullptr && err1;
        const double* base0 = leaf0 ? (fold0 ? a.PTfold : a.PT) + (size_t)key_c0 * msz : tab + tab_off0;
        const double* base1 = leaf1 ? (fold1 ? a.PTfold : a.PT) + (size_t)key_c1 * msz : tab + tab_off1;
        // a count beyond the column range: no such column (cafe/cafe_tree.c:208-209)
        const bool live = f < n_live && !(leaf0 && i0 > a.C - 1) && !(leaf1 && i1 > a.C - 1);
        const double* c0 = base0 + (size_t)(live ? i0 : 0) * a.LD;
        const double* c1 = base1 + (size_t)(live ? i1 : 0) * a.LD;
        double* L = Lbuf + (size_t)f * a.LDv;
        if (BATCH && a.LDv <= K2C_GATHER_MAX * per_state) {
            // all slices of both columns in one batch (clamped addresses, no branches)
            double g0[K2C_GATHER_MAX], g1[K2C_GATHER_MAX];
#pragma unroll
            for (int q = 0; q < K2C_GATHER_MAX; ++q) {
                const int kc = min(l + q * per_state, a.C - 1);
                g0[q] = c0[kc];
                g1[q] = c1[kc];
            }
#pragma unroll
            for (int q = 0; q < K2C_GATHER_MAX; ++q) {
                const int k = l + q * per_state;
                if (k < a.LDv) L[k] = (live && k < a.C) ? g0[q] * g1[q] : 0.0;
            }
        } else {
            for (int k = l; k < a.LDv; k += per_state) L[k] = (live && k < a.C) ? c0[k] * c1[k] : 0.0;
        }
    }
    K2C_STAMP(3);
    __syncthreads();
    K2C_STAMP(4);
    cafe_d4 fac[NFT_W][NRT_W];
#pragma unroll
    for (int i = 0; i < NFT_W; ++i)
#pragma unroll
        for (int j = 0; j < NRT_W; ++j) fac[i][j] = cafe_d4{0.0, 0.0, 0.0, 0.0};
    if (ntile > 0) {
        const double* ap = Lbuf + (size_t)li * a.LDv + lk;
        {
            unsigned voff[NRT_W];
#pragma unroll
            for (int j = 0; j < NRT_W; ++j) voff[j] = (unsigned)(lk * a.LD + li + ((j < ntile) ? (rt0 + j) : rt0) * 16) * 8u;
            if constexpr (PAIR) {
                static_assert(!PAIR || NRT_W == 2, "pairs: two row tiles per wave");
                if (ntile == 2) {
                    voff[0] = (unsigned)(lk * a.LD + 2 * li + rt0 * 16) * 8u;   // rows rt0 * 16 + 2 li, + 1: one 16-byte load
                    mfma_edge_p<NFT_W, NRT_W, 2, CAFE_K2C_DEPTH, true>(sb, voff, kstride_bytes, ap, 16 * a.LDv, a.ksteps, fac);
                } else {   // (the last wave of an odd number of row tiles)
                    mfma_edge_p<NFT_W, NRT_W, 1, CAFE_K2C_DEPTH>(sb, voff, kstride_bytes, ap, 16 * a.LDv, a.ksteps, fac);
                }
            } else if constexpr (NRT_W > 1) {
                if (ntile == NRT_W - 1)
                    mfma_edge_p<NFT_W, NRT_W, NRT_W - 1, CAFE_K2C_DEPTH>(sb, voff, kstride_bytes, ap, 16 * a.LDv, a.ksteps, fac);
                else
                    mfma_edge_p<NFT_W, NRT_W, NRT_W, CAFE_K2C_DEPTH>(sb, voff, kstride_bytes, ap, 16 * a.LDv, a.ksteps, fac);
            } else {
                mfma_edge_p<NFT_W, NRT_W, NRT_W, CAFE_K2C_DEPTH>(sb, voff, kstride_bytes, ap, 16 * a.LDv, a.ksteps, fac);
            }
        }
    }
    K2C_STAMP(5);
    double* const out = tab + out_off + (size_t)state0 * a.LD;
#pragma unroll
    for (int i = 0; i < NFT_W; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = i * 16 + lk + 4 * r;
            if (f >= n_live) continue;
            if constexpr (PAIR) {
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
#pragma unroll
            for (int j = 0; j < NRT_W; ++j) {
                if (j < ntile) {
                    const int row = (rt0 + j) * 16 + li;
                    out[(size_t)f * a.LD + row] = (row < a.C) ? fac[i][j][r] : 0.0;
                }
            }
        }
    }
    K2C_STAMP(6);
}

}  // namespace
