"""The multi-GPU exchange behind the C ABI (cafehost_init_comm, include/cafehost.h): native RCCL on the context's
stream.  A one-GPU box can only form a one-rank communicator (RCCL refuses two ranks on one device), which still
runs the whole path: communicator set-up from a unique id, sharding + automatic re-wiring on `load`, asynchronous
evaluation into the packed device buffers, ncclAllGather, polled pick-up, fixed-order sum, the report's staged
gathers.  Every result must equal the plain single-process run exactly."""
import gzip
import json
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TR = json.load(open(os.path.join(GOLD, "transcripts.json")))
CLI = os.path.join(ROOT, "cafe_amd", "bin", "cafehip")


def _run(lines, comm):
    from cafe_amd.shell import CafeShell
    sh = CafeShell(0, os.devnull)
    if comm:
        sh.init_comm(0, 1, sh.comm_unique_id())
    for l in lines:
        sh.dispatch(l)
    res = (list(sh.params), sh.score, sh.iterations, sh.evaluations)
    stats = sh.exchange_stats()
    _run.last_trace = sh.trace().copy()
    sh.close()
    return res, stats


@pytest.fixture(scope="module")
def test1_table(tmp_path_factory):
    fam = tmp_path_factory.mktemp("t1") / "test1_families.txt"
    with gzip.open(os.path.join(GOLD, "test1_families.txt.gz"), "rb") as f, open(fam, "wb") as o:
        shutil.copyfileobj(f, o)
    return str(fam)


def test_search_through_a_one_rank_rccl_communicator_equals_the_plain_run(test1_table):
    g = TR["test1"]
    lines = ["seed 10", "tree " + g["newick"], "load -i %s -max_size 20" % test1_table, "lambda -s"]
    plain, st0 = _run(lines, False)
    comm, st1 = _run(lines, True)
    assert comm == plain
    assert st0[1] == 0 and st1[1] == comm[3]          # one exchange per objective evaluation
    assert plain[0][0] == pytest.approx(g["search_result"]["lambda"], abs=2e-7)


def test_reload_rewires_the_exchange(test1_table):
    # a second, larger table after a search: the packed buffers are re-sized by `load` itself (the advisor's
    # round-1 finding: stale exchange buffers after a C++-side load)
    g = TR["test1"]
    lines = ["seed 10", "tree " + g["newick"], "load -i %s -max_size 5" % test1_table, "lambda -s",
             "load -i %s -max_size 20" % test1_table, "lambda -s"]
    plain, _ = _run(lines, False)
    tp = _run.last_trace
    comm, _ = _run(lines, True)
    tc = _run.last_trace
    import numpy as np
    d = np.nonzero((tp != tc).any(axis=1))[0] if tp.shape == tc.shape else None
    assert comm == plain, ("evaluations of the last search that differ", d, tp[d[:4]].tolist() if d is not None else None, tc[d[:4]].tolist() if d is not None else None)


def test_report_through_the_communicator_equals_the_golden_file(tmp_path):
    g = TR["test2"]
    out = str(tmp_path / "test2")
    lines = ["seed 10", "load -i %s -p 0.05 -max_size 20" % os.path.join(GOLD, "test2_families.txt"), "tree " + g["newick"],
             "lambda -s", "report " + out]
    _run(lines, True)
    got = open(out + ".cafe").read().splitlines()
    exp = open(os.path.join(GOLD, "test2.cafe")).read().splitlines()
    assert got[:1] + got[2:] == exp[:1] + exp[2:]


def test_lhtest_through_the_communicator_equals_the_plain_run(tmp_path):
    # the advisor's round-1 finding: lhtest loads a new table per file, which left the exchange wired for the
    # previous one; with the native communicator every `load` re-wires it
    newick = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"
    sim = tmp_path / "sim"
    sim.mkdir()
    base = ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + newick, "lambda -s",
            "genfamily %s/rnd -t 2" % sim]
    outs = []
    for comm in (False, True):
        out = str(tmp_path / ("lh_%d.txt" % comm))
        _run(base + ["lhtest -d %s -t (((1,1)1,(2,2)2)2,2) -l 0.0107527 -o %s" % (sim, out)], comm)
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") == 2


def test_command_line_front_end_runs_sharded_without_python(tmp_path, test1_table):
    g = TR["test1"]
    script = tmp_path / "run.sh"
    script.write_text("\n".join(["seed 10", "tree " + g["newick"], "load -i %s -max_size 20" % test1_table, "lambda -s"]) + "\n")

    def result(args):
        out = subprocess.run([CLI] + args + [str(script)], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
        assert out.returncode == 0, out.stderr[-2000:]
        m = re.findall(r"Lambda Search Result: (\d+)\s*\nLambda : (\S+) & Score: (\S+)", out.stdout)
        assert m, out.stdout[-2000:]
        return m[-1], out.stderr

    plain, _ = result([])
    comm, err = result(["--comm"])
    assert comm == plain
    assert re.search(r"1 ranks, \d+ exchanges", err)
