"""The multi-GPU search path end to end on ONE GPU: two / three ranks (gloo, sharing device 0) run the
test2/test1 scripts through cafe_amd.multi_gpu; the fitted lambda, score and evaluation count must be
IDENTICAL to the single-process run (the sharded score is bit-identical, so Nelder-Mead takes the same
decisions on every rank)."""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TR = json.load(open(os.path.join(GOLD, "transcripts.json")))


def _run_multi(script, nproc, port):
    env = dict(os.environ, CAFE_BACKEND="gloo", CAFE_SAME_DEVICE="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "cafe_amd.multi_gpu", script]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"params \[(.*?)\] score (\S+) iterations (\d+) evaluations (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    params = [float(re.sub(r"np\.float64\(|\)", "", x)) for x in m.group(1).split(",")]
    return params, float(m.group(2)), int(m.group(3)), int(m.group(4))


def _run_single(lines):
    from cafe_amd.shell import CafeShell
    sh = CafeShell(0, os.devnull)
    for l in lines:
        sh.dispatch(l)
    res = (list(sh.params), sh.score, sh.iterations, sh.evaluations)
    sh.close()
    return res


@pytest.mark.parametrize("nproc", [2, 3])
def test_sharded_search_equals_single_gpu(tmp_path, nproc):
    g = TR["test1"]
    fam = tmp_path / "test1_families.txt"
    with gzip.open(os.path.join(GOLD, "test1_families.txt.gz"), "rb") as f, open(fam, "wb") as o:
        shutil.copyfileobj(f, o)
    lines = ["seed 10", "tree " + g["newick"], "load -i %s -max_size 20" % fam, "lambda -s"]
    script = tmp_path / "run.sh"
    script.write_text("\n".join(lines) + "\n")
    p1, s1, it1, ev1 = _run_single(lines)
    pm, sm, itm, evm = _run_multi(str(script), nproc, 29620 + nproc)
    assert pm == p1 and sm == s1 and itm == it1 and evm == ev1
    assert p1[0] == pytest.approx(g["search_result"]["lambda"], abs=2e-7)


@pytest.mark.parametrize("nproc", [2, 3])
def test_sharded_report_equals_golden_test2_cafe(tmp_path, nproc):
    # `report` on several ranks: the Monte-Carlo null is sharded by root size, the observed families by block,
    # rank 0 writes the file -- which must still be the reference's golden text (seed 10, -t 1 draw order)
    g = TR["test2"]
    out = str(tmp_path / "test2")
    lines = ["seed 10", "load -i %s -p 0.05 -max_size 20" % os.path.join(GOLD, "test2_families.txt"),
             "tree " + g["newick"], "lambda -s", "report " + out, "pvalue -o " + out + ".pv"]
    script = tmp_path / "run.sh"
    script.write_text("\n".join(lines) + "\n")
    _run_multi(str(script), nproc, 29640 + nproc)
    got = open(out + ".cafe").read().splitlines()
    exp = open(os.path.join(GOLD, "test2.cafe")).read().splitlines()
    assert exp[1].endswith(got[1]) and got[1] == "Lambda:\t0.00133949"
    assert got[:1] + got[2:] == exp[:1] + exp[2:]
    # the conditional distribution written by `pvalue -o`: root sizes x 1000 sorted likelihoods
    rows = open(out + ".pv").read().splitlines()
    assert len(rows) == 30 and all(len(r.split("\t")) == 1000 for r in rows)


def test_sharded_error_model_pipeline_equals_single_process(tmp_path):
    # BASELINE configs[4] in miniature: error model on every leaf, lambda -s, report (Monte-Carlo null sharded by
    # root size, 59 observed families sharded by block) on 3 ranks -- fitted lambda, score and the whole report
    # text must equal the single-process run
    newick = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"
    out_m, out_s = str(tmp_path / "multi"), str(tmp_path / "single")
    common = ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + newick,
              "errormodel -model %s -all" % os.path.join(GOLD, "error1.txt"), "lambda -s"]
    script = tmp_path / "run.sh"
    script.write_text("\n".join(common + ["report " + out_m]) + "\n")
    p1, s1, it1, ev1 = _run_single(common + ["report " + out_s])
    pm, sm, itm, evm = _run_multi(str(script), 3, 29655)
    assert pm == p1 and sm == s1 and itm == it1 and evm == ev1
    a = open(out_m + ".cafe").read()
    b = open(out_s + ".cafe").read()
    assert a == b and a.count("\n") > 60
