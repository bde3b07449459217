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


def _ranks(tmp_path, world, mode, **env):
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    procs = []
    for r in range(world):
        out = tmp_path / ("r%d.json" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "gpu", str(r), str(world),
                                             str(idfile), str(out), CFG, str(F_TOTAL), mode, str(STEPS)], cwd=ROOT,
                                            env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="90", **env))))
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
        st = r["status"]
        # the functional probe heard every peer on every rank; RCCL was never loaded and the record says so
        assert st["mode_agreed_at_init"] == "direct" and st["peers_heard_by_probe"] == world and st["peers_mapped"] == world
        assert st["comm_world"] == world and not st["rccl_initialised"] and st["rccl_ranks"] == 0
        assert r["init_seconds"] < 30


def test_a_mapped_but_unreachable_peer_sends_every_rank_the_same_way(tmp_path, single):
    # rank 1 maps its peers but its probe stores never leave (CAFEHIP_COMM_INJECT=mute:1): rank 0 does not hear it, the
    # ranks agree -- at set-up, within seconds -- to leave the direct mode TOGETHER.  Two ranks on one device cannot form an
    # RCCL communicator, so here "together" normally means every rank's cafehip_comm_init fails with the same reason;
    # where RCCL does accept them, every rank runs in rccl mode with the single-context bits.
    res = _ranks(tmp_path, 2, "auto", CAFEHIP_COMM_INJECT="mute:1")
    failed = ["init_error" in r for r in res]
    assert failed[0] == failed[1], res
    for r in res:
        assert r["init_seconds"] < 60, r
    if failed[0]:
        for r in res:
            assert "no exchange mode works on every rank" in r["init_error"], r
    else:
        for r in res:
            assert r["status"]["mode"] == "rccl" and r["status"]["rccl_ranks"] == 2 and not r["status"]["direct_ok_on_every_rank"]
            assert [tuple(x) for x in r["scores"]] == single


def test_a_late_peer_is_waited_for_in_slices_not_inside_one_kernel(tmp_path, single):
    # rank 1 arrives 3.5 s late for the second evaluation: rank 0's score kernel gives up after its ~1 s slice, the host
    # re-polls with the one-workgroup wait kernel, and the evaluation completes with the same bits
    res = _ranks(tmp_path, 2, "direct", COMM_WORKER_DELAY="1:1:3.5")
    for r in res:
        assert [tuple(x) for x in r["scores"]] == single, r["rank"]
    assert res[0]["status"]["host_paced_repolls"] >= 2 and res[1]["status"]["host_paced_repolls"] == 0


def test_ranks_out_of_step_are_realigned_by_a_collective_resync(tmp_path, single):
    # rank 1 leaves the second evaluation out: rank 0's call fails after its patience (4 s here: four 1 s slices, the GPU is
    # never inside one kernel for longer) instead of hanging; cafehip_comm_resync on both ranks clears the exchange buffers
    # and restarts the sequence numbers, and every later evaluation carries the single-context bits again
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    procs = []
    for r in range(2):
        out = tmp_path / ("r%d.json" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "gpu", str(r), "2", str(idfile), str(out),
                                             CFG, str(F_TOTAL), "direct", str(STEPS)], cwd=ROOT,
                                            env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="4", COMM_WORKER_SKIP="1:1"))))
    res = []
    for out, p in procs:
        assert p.wait(timeout=600) == 0
        res.append(json.load(open(out)))
    assert len(res[0]["failures"]) == 1 and "did not deliver its row" in res[0]["failures"][0], res[0]["failures"]
    assert res[1]["failures"] == []
    for r in res:
        got = [tuple(x) for x in r["scores"]]
        assert got[1] == ("skipped", -2)
        assert got[:1] + got[2:] == single[:1] + single[2:], r["rank"]


def _device_count():
    import ctypes as C
    n = C.c_int(0)
    try:
        hip = C.CDLL("libamdhip64.so")
        if hip.hipGetDeviceCount(C.byref(n)) != 0:
            return 0
    except OSError:
        return 0
    return n.value


@pytest.mark.parametrize("mode", ["direct", "rccl"])
def test_ranks_on_distinct_devices_are_bit_identical_to_one_context(tmp_path, single, mode):
    # the real thing: rank r on device r (needs a box with >= 2 GPUs; the builder's boxes have one, so this has only ever
    # been skipped there -- it is the first test to run when a multi-GPU node sees the suite)
    n = _device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % n)
    world = min(n, 4)
    res = _ranks(tmp_path, world, mode, COMM_WORKER_DEVICE_PER_RANK="1")
    for r in res:
        assert "init_error" not in r, r
        assert [tuple(x) for x in r["scores"]] == single, (world, r["rank"])
        assert r["status"]["mode"] == mode
        if mode == "rccl":
            assert r["status"]["rccl_initialised"] and r["status"]["rccl_ranks"] == world


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


def test_lhtest_files_are_dealt_to_the_ranks_and_the_output_is_the_one_rank_file(tmp_path):
    # SURVEY.md 8 f-4 / cafe/cafe_commands.cpp:1473-1536: lhtest is 7 independent two-search pipelines on small simulated
    # tables.  Sharded, rank r runs files r, r + N, ... WHOLE on its GPU; rank 0 writes the gathered lines in file order --
    # byte for byte the one-rank file, and the session ends where the one-rank run ends (the `lambda -s` after it draws the
    # same random start on the same table: same result line).  CAFEHOST_LHTEST_DEAL=0 is the round-3 behaviour (every
    # rank runs every search on its block of every table): same bytes again.
    import time
    GOLD = os.path.join(ROOT, "tests", "golden")
    newick = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"
    sim = tmp_path / "sim"
    sim.mkdir()
    gen = tmp_path / "gen.sh"
    gen.write_text("\n".join(["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + newick, "lambda -s",
                              "genfamily %s/rnd -t 7" % sim]) + "\n")
    r = subprocess.run([CLI, str(gen)], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([f for f in os.listdir(sim) if f.endswith(".tab")]) == 7

    def run(world, deal=True):
        out = tmp_path / ("lh_%d_%d.out" % (world, deal))
        script = tmp_path / ("lh_%d_%d.sh" % (world, deal))
        script.write_text("\n".join(["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + newick,
                                     "lambda -s", "lhtest -d %s -t (((1,1)1,(2,2)2)2,2) -l 0.0107527 -o %s" % (sim, out),
                                     "lambda -s"]) + "\n")
        args = [CLI] + (["--gpus", str(world), "--same-device"] if world > 1 else []) + [str(script)]
        t0 = time.time()
        r = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=str(tmp_path),
                           env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="90", CAFEHOST_LHTEST_DEAL="1" if deal else "0"))
        assert r.returncode == 0, r.stderr[-2000:]
        last = re.findall(r"Lambda Search Result: (\d+)\s*\nLambda : (\S+) & Score: (\S+)", r.stdout)[-1]
        return open(out).read(), last, time.time() - t0

    one, last_one, t_one = run(1)
    assert one.count("\n") == 7
    times = {1: t_one}
    for world in (2, 3):
        got, last, t = run(world)
        times[world] = t
        assert got == one, world
        assert last == last_one, world      # the session after lhtest is the one-rank session: table, model, random stream
    got, last, t = run(2, deal=False)
    assert got == one and last == last_one
    print("lhtest wall clock, 7 files, ranks sharing ONE device: %s; undealt 2 ranks %.2f s" % (times, t))


def test_lambda_grid_points_are_dealt_to_the_ranks_and_the_surface_is_the_one_rank_file(tmp_path):
    # cafe/lambda.cpp:192-231 (cafe_lambda_distribution): `lambda -r a:b:c` evaluates a grid of independent points.  In a
    # sharded job on a table that does not fill a GPU, rank r evaluates points r, r + N, ... on the WHOLE table and the values
    # are gathered (option grid_deal, CAFEHOST_GRID_DEAL): the log lines and the -o file must be the one-rank run's, byte for
    # byte -- also with the dealing switched off (every rank evaluates every point on its block), and the session after it
    # must be the one-rank session (the search that follows draws the same start on the same table).
    GOLD = os.path.join(ROOT, "tests", "golden")
    newick = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"

    def run(world, deal):
        out = tmp_path / ("grid_%d_%d.txt" % (world, deal))
        script = tmp_path / ("grid_%d_%d.sh" % (world, deal))
        script.write_text("\n".join(["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + newick,
                                     "lambda -r 0.001:0.0015:0.0175 -o %s" % out, "lambda -s"]) + "\n")
        args = [CLI] + (["--gpus", str(world), "--same-device"] if world > 1 else []) + [str(script)]
        r = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=str(tmp_path),
                           env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="90", CAFEHOST_GRID_DEAL=str(deal), CAFEHOST_TIMING="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = re.findall(r"Lambda : \S+ & Score: \S+", r.stdout)
        return open(out).read(), lines, ("grid points dealt" in r.stderr)

    surface, lines, dealt = run(1, 1)
    assert surface.count("\n") == 12 and len(lines) > 12 and not dealt
    for world in (2, 3):
        got, ln, dealt = run(world, 1)
        assert dealt, world
        assert got == surface and ln == lines, world
    got, ln, dealt = run(2, 0)
    assert not dealt and got == surface and ln == lines
