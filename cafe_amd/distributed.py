"""Family sharding and the single exchange step of the multi-GPU path (one process per GPU).

Families are independent given the transition matrices (cafe/lambda.cpp:698-722 is a map + sum), so
each rank scores a contiguous block of the count table and the ranks exchange only the per-chunk
partial sums of log max-posterior (CAFEHIP_CHUNK = 256 families per chunk, in family order) and the
index of the first zero-likelihood family.  Blocks are chunk-aligned and the final sum runs over the
gathered chunk list in chunk order on every rank, so the score is bit-identical for any number of
ranks.

Backend-agnostic: "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests.
"""
import math

import numpy as np

CHUNK = 256
NO_ZERO = 2 ** 31 - 1


def shard_bounds(F, world, chunk=CHUNK):
    """Contiguous chunk-aligned blocks: [(lo, hi)] per rank covering [0, F)."""
    n_chunks = (F + chunk - 1) // chunk
    base, extra = divmod(n_chunks, world)
    out = []
    c0 = 0
    for r in range(world):
        nc = base + (1 if r < extra else 0)
        lo = min(c0 * chunk, F)
        hi = min((c0 + nc) * chunk, F)
        out.append((lo, hi))
        c0 += nc
    return out


def max_chunks_per_rank(F, world, chunk=CHUNK):
    return max((hi - lo + chunk - 1) // chunk for lo, hi in shard_bounds(F, world, chunk))


def chunk_tree_sums(values, chunk=CHUNK):
    """Per-chunk sums with the fixed halving tree of the k3_score kernel (cafe_amd/csrc/cafehip.hip):
    red[t] += red[t + s] for s = chunk/2 ... 1.  Host mirror used by the CPU tests."""
    v = np.asarray(values, np.float64)
    n_chunks = (len(v) + chunk - 1) // chunk
    pad = np.zeros(n_chunks * chunk)
    pad[:len(v)] = v
    red = pad.reshape(n_chunks, chunk).copy()
    s = chunk // 2
    while s > 0:
        red[:, :s] += red[:, s:2 * s]
        s //= 2
    return red[:, 0].copy()


def final_score(all_chunk_sums, first_zero):
    """Fixed-order sum over the gathered chunk list; -inf when any family has likelihood 0
    (cafe/lambda.cpp:715-720, 753-760)."""
    if first_zero != NO_ZERO:
        return -math.inf
    a = np.asarray(all_chunk_sums, np.float64)
    if a.size == 0:
        return 0.0
    # strictly left-to-right (ufunc.accumulate is a plain sequential loop, unlike the pairwise add.reduce): the
    # same additions as the C loop of the single-GPU path, without a Python-level loop per chunk
    return float(np.add.accumulate(a)[-1])


def exchange(dist, torch, chunk_sums, first_zero_local, lo, n_local, slots, device):
    """all_gather the (zero-padded) chunk sums, all_reduce(min) the global first-zero index.
    chunk_sums: 1-D float64 tensor of this rank's chunks; first_zero_local: int tensor[1] holding a
    LOCAL index or any value >= n_local for none.  Returns (host array of world*slots sums, fz)."""
    world = dist.get_world_size()
    padded = torch.zeros(slots, dtype=torch.float64, device=device)
    padded[:chunk_sums.numel()] = chunk_sums
    gathered = torch.zeros(slots * world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, padded)
    fz = first_zero_local.to(torch.int64)
    fzg = torch.where(fz < n_local, fz + lo, torch.full_like(fz, NO_ZERO))
    dist.all_reduce(fzg, op=dist.ReduceOp.MIN)
    return gathered.cpu().numpy(), int(fzg.item())


def packed_buffer(torch, slots, device):
    """One float64 tensor per rank: [0, slots) chunk sums (unused slots stay 0), element `slots` holds
    the LOCAL first-zero index as an int32 in its low 4 bytes (written by the k3 kernel through the
    d_first_zero pointer).  Returns (tensor, chunk_sums_ptr, first_zero_ptr)."""
    t = torch.zeros(slots + 1, dtype=torch.float64, device=device)
    return t, t.data_ptr(), t.data_ptr() + 8 * slots


def exchange_packed(dist, torch, packed, gathered, slots, bounds, host_pinned=None, engine=None):
    """The whole exchange step as ONE collective: all_gather of the packed per-rank buffers, then on the
    host the fixed-order sum over chunks and the global first-zero index.  bounds[r] = (lo, hi) of rank r.
    host_pinned: optional pinned CPU tensor of gathered's shape (one async copy + stream sync instead of a
    pageable .cpu())."""
    dist.all_gather_into_tensor(gathered, packed)
    if engine is not None:
        # the collective ran on the engine's stream: a one-workgroup kernel behind it writes the result into
        # pinned host memory and a flag the host polls (no copy command, no stream synchronisation)
        host = engine.fetch_small(gathered.data_ptr(), gathered.numel())
    elif host_pinned is not None:
        host_pinned.copy_(gathered, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = host_pinned.numpy()
    else:
        host = gathered.cpu().numpy()                  # the optimiser needs the value on the host
    world = len(bounds)
    rows = host.reshape(world, slots + 1)
    fz_local = rows[:, slots].copy().view(np.int32)[0::2]
    fz = NO_ZERO
    for r, (lo, hi) in enumerate(bounds):
        if 0 <= fz_local[r] < hi - lo:
            fz = min(fz, lo + int(fz_local[r]))
    return final_score(rows[:, :slots].reshape(-1), fz), fz
