// k3_device.hpp -- the body of the score kernel as a device function of (block, grid), so that it can run as the leading
// blocks of a launch that does something else behind them (k1_matrices.hip: score + matrices of the next candidates, round 5).
// == get_posterior's sum and first zero-likelihood family, cafe/lambda.cpp:691-724.
#pragma once
#include <climits>
#include <cmath>

#include "device_types.hpp"

namespace cafehip {

// (bx, gx): chunk and chunk count; (by, gy): parameter set and set count.  blockDim.x == CAFEHIP_CHUNK.
template <bool HOST_OUT>
__device__ __forceinline__ void k3_score_block(const K3Args& a, const int bx, const int gx, const int by, const int gy)
{
    const double* __restrict__ max_post_u = a.max_post_u;
    const double* __restrict__ max_lik_u = a.max_lik_u;
    const int32_t* __restrict__ fam2u = a.fam2u;
    const int F = a.F, Fu = a.Fu;
    double* __restrict__ chunk_sums = a.chunk_sums;
    int32_t* __restrict__ first_zero = a.first_zero;
    HostResult* host = a.host;
    int32_t* arrive = a.arrive;
    const int32_t seq = a.seq;
    // by = parameter set: its per-family values start at set * Fu, its chunk sums at set * gx
    __shared__ double red[CAFEHIP_CHUNK];
    __shared__ int s_last;
    const int set = by;
    max_post_u += (size_t)set * Fu;
    max_lik_u += (size_t)set * Fu;
    const int i = bx * CAFEHIP_CHUNK + threadIdx.x;
    double v = 0.0;
    if (i < F) {
        const int u = fam2u ? fam2u[i] : i;   // (NULL: no duplicate rows, family i is unique row i -- one round trip less)
        v = log(max_post_u[u]);                                   // cafe/lambda.cpp:721
        if (max_lik_u[u] == 0.0) atomicMin(first_zero + set, i);  // cafe/lambda.cpp:715-720
    }
    red[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = CAFEHIP_CHUNK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const size_t slot = (size_t)set * gx + bx;
    if (!HOST_OUT) {
        if (threadIdx.x == 0) chunk_sums[slot] = red[0];
        return;
    }
    if (threadIdx.x == 0) {
        host->chunk_sums[slot] = red[0];
        __threadfence_system();
        s_last = (atomicAdd(arrive, 1) == gx * gy - 1);
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        // (every other block's chunk sum was fenced system-wide before it counted itself in)
        // atomic read of the final value, which also leaves the word as the NEXT evaluation needs it (INT32_MAX): an
        // evaluation whose matrices are already on the device (cafehip_prefetch_matrices) has no K1 launch to reset it
        const int32_t fz0 = atomicExch(first_zero, INT32_MAX);
        if (gy > 1) {
            for (int q = 1; q < gy; ++q) host->first_zero[q] = atomicExch(first_zero + q, INT32_MAX);
            __threadfence_system();
        }
        *arrive = 0;
        // the sequence number and set 0's first-zero index share one aligned 8-byte word: a single store publishes
        // both, no fence in between (the host reads the index after it has seen the number)
        static_assert(offsetof(HostResult, first_zero) == 4 && offsetof(HostResult, done_seq) == 0, "one 8-byte word");
        *reinterpret_cast<volatile unsigned long long*>(host) = ((unsigned long long)(unsigned)fz0 << 32) | (unsigned)seq;
    }
}

}  // namespace cafehip
