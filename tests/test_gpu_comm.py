"""The multi-GPU exchange behind the kernel library's ABI (cafehip_comm_*, include/cafehip.h), on ONE GPU: 1, 2 and 3
processes share device 0, each holding its chunk-aligned block of the table, and evaluate through
cafehip_eval_posterior_sharded with the DIRECT exchange (every rank's score kernel stores its packed row into the
other ranks' hipIpc-mapped buffers and waits for theirs).  Scores and first-zero indices must equal the single-context
evaluation of the whole table BIT FOR BIT, on every rank, for every number of ranks; a second table through the same
communicator re-wires the exchange.  RCCL refuses two ranks on one device, so the rccl mode runs with one rank only
(same bits again).  The host driver's sharded commands ride on the same calls (tests/test_gpu_native_comm.py)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cafe_amd", "bin", "cafehip")
CFG, F_TOTAL, STEPS = "cfg2", 2900, 4


@pytest.fixture(scope="module")
def single():
    """The same evaluations on one context holding the whole table."""
    import cafe_amd
    from cafe_amd import distributed as D
    from cafe_amd import prior as cprior
    from cafe_amd import synth
    from cafe_amd import tree as ctree
    cfg = dict(synth.CONFIGS[CFG])
    tree = ctree.CafeTree(synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"])))
    counts = synth.simulate_families(tree, F_TOTAL, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    rng = cafe_amd.init_family_size(cfg["m"])
    prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    out = []
    for s in range(STEPS):
        nl, nm = synth.node_rates(tree, cfg, 1.0 + 0.01 * s, 1.0 + 0.007 * s)
        if s == STEPS - 1:
            nl = nl * 400.0
        sc, fz = eng.get_posterior(nl, nm, prior)
        out.append((float(sc).hex(), int(fz)))
    half = max(D.CHUNK, (F_TOTAL // 2 // D.CHUNK) * D.CHUNK)
    eng.set_families(counts[:half], rng)
    nl, nm = synth.node_rates(tree, cfg)
    sc, fz = eng.get_posterior(nl, nm, prior)
    out.append((float(sc).hex(), int(fz)))
    eng.close()
    assert out[STEPS - 1][1] >= 0 and out[0][1] == -1    # the absurd rates do produce a zero-likelihood family
    return out


def _ranks(tmp_path, world, mode):
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    procs = []
    for r in range(world):
        out = tmp_path / ("r%d.json" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "gpu", str(r), str(world),
                                             str(idfile), str(out), CFG, str(F_TOTAL), mode, str(STEPS)], cwd=ROOT,
                                            env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="90"))))
    res = []
    for out, p in procs:
        assert p.wait(timeout=600) == 0
        res.append(json.load(open(out)))
    return res


@pytest.mark.parametrize("world", [1, 2, 3])
def test_direct_exchange_is_bit_identical_to_one_context(tmp_path, single, world):
    res = _ranks(tmp_path, world, "direct")
    for r in res:
        assert [tuple(x) for x in r["scores"]] == single, (world, r["rank"])
        assert r["info"]["mode"] == "direct" and r["info"]["world"] == world and r["info"]["calls"] == STEPS + 1


def test_rccl_exchange_with_one_rank_is_bit_identical(tmp_path, single):
    res = _ranks(tmp_path, 1, "rccl")
    assert [tuple(x) for x in res[0]["scores"]] == single
    assert res[0]["info"]["mode"] == "rccl"


@pytest.mark.parametrize("world", [2, 3])
def test_command_line_front_end_shards_over_ranks_of_one_device(tmp_path, world):
    # cafehip --gpus N --same-device: N processes, no Python, the native exchange; search AND report
    import gzip
    import shutil
    GOLD = os.path.join(ROOT, "tests", "golden")
    TR = json.load(open(os.path.join(GOLD, "transcripts.json")))
    g = TR["test2"]
    out = str(tmp_path / "test2")
    script = tmp_path / "run.sh"
    script.write_text("\n".join(["seed 10", "load -i %s -p 0.05 -max_size 20" % os.path.join(GOLD, "test2_families.txt"),
                                 "tree " + g["newick"], "lambda -s", "report " + out]) + "\n")
    r = subprocess.run([CLI, "--gpus", str(world), "--same-device", str(script)], capture_output=True, text=True, timeout=900,
                       cwd=str(tmp_path), env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="90"))
    assert r.returncode == 0, r.stderr[-2000:]
    got = open(out + ".cafe").read().splitlines()
    exp = open(os.path.join(GOLD, "test2.cafe")).read().splitlines()
    assert got[:1] + got[2:] == exp[:1] + exp[2:]
    assert re.search(r"%d ranks, \d+ exchanges" % world, r.stderr)
