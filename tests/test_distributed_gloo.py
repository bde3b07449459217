"""world_size-2 (and 3) gloo test of the multi-GPU path's host logic on CPU: chunk-aligned
sharding, zero-padded all_gather of chunk sums, all_reduce(min) of the first-zero index and the
fixed-order final sum give the SAME bits as a single rank.  The per-shard evaluator here is the
oracle (the GPU evaluator is covered by the -m gpu tests)."""
import math
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, F, lam_v, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from cafe_amd import distributed as D
    from tests import _orc as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, counts, rng, prior = _problem(F)
    lo, hi = D.shard_bounds(F, world)[rank]
    lam = np.full(t.n_nodes, lam_v)
    mu = np.full(t.n_nodes, -1.0)
    if hi > lo:
        score, fz, ml, am, mp = O.eval_posterior(t, counts[lo:hi], rng, lam, mu, prior)
        logs = np.log(mp)
        sums = D.chunk_tree_sums(logs)
        fz_local = fz if fz >= 0 else hi - lo
    else:
        sums = np.zeros(0)
        fz_local = 0
    slots = D.max_chunks_per_rank(F, world)
    all_sums, fzg = D.exchange(dist, torch, torch.from_numpy(sums), torch.tensor([fz_local], dtype=torch.int32), lo,
                               hi - lo, slots, "cpu")
    # the packed single-collective variant used by bench.py must agree
    packed, _, _ = D.packed_buffer(torch, slots, "cpu")
    packed[:len(sums)] = torch.from_numpy(sums)
    packed[slots:].view(torch.int32)[0] = int(fz_local)
    gathered = torch.zeros((slots + 1) * world, dtype=torch.float64)
    score2, fz2 = D.exchange_packed(dist, torch, packed, gathered, slots, D.shard_bounds(F, world))
    assert fz2 == fzg and (score2 == D.final_score(all_sums, fzg))
    q.put((rank, D.final_score(all_sums, fzg), fzg))
    dist.destroy_process_group()


def _problem(F):
    from tests import _orc as O
    t = O.PyTree("(((A:6,B:6):81,(C:17,D:17):70):6,E:93)")
    rs = np.random.RandomState(11)
    counts = rs.poisson(4, size=(F, 5)).astype(np.int32)
    counts[min(700, F - 1)] = 0
    rng = O.range_from_max(int(counts.max()))
    prior = O.prior_poisson(1000, rng.root_min, 3.0)
    return t, counts, rng, prior


def _single(F, lam_v):
    from cafe_amd import distributed as D
    from tests import _orc as O
    t, counts, rng, prior = _problem(F)
    lam = np.full(t.n_nodes, lam_v)
    mu = np.full(t.n_nodes, -1.0)
    score, fz, ml, am, mp = O.eval_posterior(t, counts, rng, lam, mu, prior)
    sums = D.chunk_tree_sums(np.log(mp))
    return D.final_score(sums, fz if fz >= 0 else D.NO_ZERO), fz


@pytest.mark.parametrize("world,F", [(2, 1000), (2, 300), (3, 1500)])
def test_sharded_score_is_bit_identical(world, F):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, 0.002, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref, fz = _single(F, 0.002)
    assert fz == -1 and math.isfinite(ref)
    for rank, score, fzg in res:
        assert score == ref  # same bits on every rank, for any world size


def test_first_zero_family_is_global_minimum():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    F = 1000
    # lambda * 93 > 1 -> zero matrix on E's edge -> every family has likelihood 0
    procs = [ctx.Process(target=_worker, args=(r, 2, port, F, 0.02, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    ref, fz = _single(F, 0.02)
    assert ref == -math.inf and fz == 0
    for rank, score, fzg in res:
        assert score == -math.inf and fzg == 0


def test_shard_bounds_are_chunk_aligned_and_cover():
    from cafe_amd import distributed as D
    for F in (0, 1, 255, 256, 257, 1000, 10000, 500000):
        for world in (1, 2, 3, 4, 8):
            b = D.shard_bounds(F, world)
            assert b[0][0] == 0 and b[-1][1] == F
            for (lo, hi), (lo2, hi2) in zip(b, b[1:]):
                assert hi == lo2
            for lo, hi in b:
                assert lo % D.CHUNK == 0 or lo == F
            assert sum(hi - lo for lo, hi in b) == F


def test_chunk_tree_sum_matches_definition():
    from cafe_amd import distributed as D
    rs = np.random.RandomState(0)
    v = rs.randn(700)
    got = D.chunk_tree_sums(v)
    assert len(got) == 3
    exp = []
    for c in range(3):
        red = np.zeros(256)
        seg = v[c * 256:(c + 1) * 256]
        red[:len(seg)] = seg
        s = 128
        while s:
            for t in range(s):
                red[t] += red[t + s]
            s //= 2
        exp.append(red[0])
    assert np.array_equal(got, np.array(exp))
