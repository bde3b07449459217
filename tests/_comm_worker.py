"""One rank of the native communicator tests (started by tests/test_comm_host.py and tests/test_gpu_comm.py):
    python tests/_comm_worker.py host  <rank> <world> <idfile> <out.json>
    python tests/_comm_worker.py mode  <rank> <world> <idfile> <out.json> <probe_ok> <rccl_ok>
    python tests/_comm_worker.py oversize <rank> <world> <idfile> <out.json> <bad_rank>
    python tests/_comm_worker.py gpu   <rank> <world> <idfile> <out.json> <cfg> <F_total> <mode> <steps>
`host`: rendezvous + barriers + host all-gather through the shared-memory segment, no GPU.
`gpu`: every rank on device 0 builds its block of a synthetic table, evaluates `steps` parameter sets through
cafehip_eval_posterior_sharded and writes the scores (as hex floats: bit-exact comparison)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host(rank, world, idfile, out):
    from cafe_amd import _lib
    L = _lib.load()
    uid = open(idfile, "rb").read()
    n = 1000 + 37 * rank           # ragged blocks in fixed slots
    slot = 1000 + 37 * (world - 1)
    mine = (np.arange(n, dtype=np.uint8) * (rank + 3)).astype(np.uint8)
    allb = np.zeros(slot * world, np.uint8)
    rc = L.cafehip_comm_host_selftest(rank, world, C.c_char_p(uid), mine.ctypes.data_as(C.c_void_p), n,
                                      allb.ctypes.data_as(C.c_void_p), slot)
    ok = rc == 0
    if ok:
        for r in range(world):
            exp = (np.arange(1000 + 37 * r, dtype=np.uint8) * (r + 3)).astype(np.uint8)
            ok = ok and np.array_equal(allb[r * slot:r * slot + len(exp)], exp)
    json.dump({"rank": rank, "ok": bool(ok), "err": "" if rc == 0 else L.cafehip_last_error().decode()}, open(out, "w"))


def mode(rank, world, idfile, out, probe_ok, rccl_ok):
    """The mode agreement of cafehip_comm_init alone, this rank's local outcomes injected (no device)."""
    from cafe_amd import _lib
    L = _lib.load()
    uid = open(idfile, "rb").read()
    m = C.c_int(-7)
    rc = L.cafehip_comm_mode_selftest(rank, world, C.c_char_p(uid), int(probe_ok), int(rccl_ok), C.byref(m))
    json.dump({"rank": rank, "rc": rc, "mode": m.value, "err": "" if rc == 0 else L.cafehip_last_error().decode()}, open(out, "w"))


def oversize(rank, world, idfile, out, bad_rank):
    """Host all-gather in which `bad_rank` offers more bytes than a slot holds: it must fail THERE with the size message
    and on the other ranks at once with that rank's message, not after a barrier time-out."""
    import time
    from cafe_amd import _lib
    L = _lib.load()
    uid = open(idfile, "rb").read()
    slot = 64
    n = 200 if rank == bad_rank else 64
    mine = np.zeros(n, np.uint8)
    allb = np.zeros(slot * world, np.uint8)
    t0 = time.time()
    rc = L.cafehip_comm_host_selftest(rank, world, C.c_char_p(uid), mine.ctypes.data_as(C.c_void_p), n,
                                      allb.ctypes.data_as(C.c_void_p), slot)
    json.dump({"rank": rank, "rc": rc, "seconds": time.time() - t0, "err": "" if rc == 0 else L.cafehip_last_error().decode()}, open(out, "w"))


def gpu(rank, world, idfile, out, cfg_name, F_total, mode, steps):
    import cafe_amd
    from cafe_amd import distributed as D
    from cafe_amd import prior as cprior
    from cafe_amd import synth
    from cafe_amd import tree as ctree
    cfg = dict(synth.CONFIGS[cfg_name])
    tree = ctree.CafeTree(synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"])))
    counts = synth.simulate_families(tree, F_total, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    rng = cafe_amd.init_family_size(cfg["m"])
    prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
    bounds = D.shard_bounds(F_total, world)
    lo, hi = bounds[rank]
    import time
    eng = cafe_amd.Engine(rank if os.environ.get("COMM_WORKER_DEVICE_PER_RANK") else 0)
    eng.set_option("comm", mode)
    t0 = time.time()
    try:
        eng.comm_init(rank, world, open(idfile, "rb").read())
    except Exception as e:   # noqa: BLE001 -- the test wants to see what every rank was told
        json.dump({"rank": rank, "init_error": str(e), "init_seconds": time.time() - t0}, open(out, "w"))
        eng.close()
        return
    init_seconds = time.time() - t0
    delay = os.environ.get("COMM_WORKER_DELAY", "")   # "rank:step:seconds": that rank is late for that evaluation
    d_rank, d_step, d_sec = (int(delay.split(":")[0]), int(delay.split(":")[1]), float(delay.split(":")[2])) if delay else (-1, -1, 0.0)
    skip = os.environ.get("COMM_WORKER_SKIP", "")     # "rank:step": that rank leaves one evaluation out (its peers time out)
    s_rank, s_step = (int(skip.split(":")[0]), int(skip.split(":")[1])) if skip else (-1, -1)
    failures = []
    tree.apply(eng)
    eng.set_families(counts[lo:hi], rng)
    eng.comm_set_blocks(bounds)
    scores = []
    for s in range(steps):
        nl, nm = synth.node_rates(tree, cfg, 1.0 + 0.01 * s, 1.0 + 0.007 * s)
        if s == steps - 1:
            nl = nl * 400.0        # absurd rates: some family gets zero likelihood -> -inf and a first-zero index
        if rank == d_rank and s == d_step:
            time.sleep(d_sec)
        if s == s_step and s_rank >= 0:
            # one rank skips this evaluation: the others' call must FAIL (not hang), and a collective resync must put
            # every rank back in step (the sequence numbers differ by one from here on otherwise: ADVICE r03)
            if rank == s_rank:
                time.sleep(3.0)
            else:
                try:
                    eng.get_posterior_sharded(nl, nm, prior)
                    failures.append("no error")
                except Exception as e:   # noqa: BLE001
                    failures.append(str(e))
            eng.comm_resync()
            scores.append(("skipped", -2))
            continue
        sc, fz = eng.get_posterior_sharded(nl, nm, prior)
        scores.append((float(sc).hex(), int(fz)))
    # a second table through the same communicator (re-wiring)
    half = max(D.CHUNK, (F_total // 2 // D.CHUNK) * D.CHUNK)
    b2 = D.shard_bounds(half, world)
    eng.set_families(counts[b2[rank][0]:b2[rank][1]], rng)
    eng.comm_set_blocks(b2)
    nl, nm = synth.node_rates(tree, cfg)
    sc, fz = eng.get_posterior_sharded(nl, nm, prior)
    scores.append((float(sc).hex(), int(fz)))
    info = eng.comm_info()
    status = eng.comm_status()
    eng.close()
    json.dump({"rank": rank, "scores": scores, "info": info, "status": status, "init_seconds": init_seconds, "failures": failures}, open(out, "w"))


if __name__ == "__main__":
    kind, rank, world, idfile, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    if kind == "host":
        host(rank, world, idfile, out)
    elif kind == "mode":
        mode(rank, world, idfile, out, int(sys.argv[6]), int(sys.argv[7]))
    elif kind == "oversize":
        oversize(rank, world, idfile, out, int(sys.argv[6]))
    else:
        gpu(rank, world, idfile, out, sys.argv[6], int(sys.argv[7]), sys.argv[8], int(sys.argv[9]))
