"""The reference's third golden transcript (tests/integration/test3.t): a two-class lambda search on the example
table, then rootdist + genfamily (10 simulated tables of 11,433 families drawn with glibc rand()) + lhtest
(per table: a global-lambda search and a two-class search).  Every printed number of the transcript -- table
sizes and ranges of the SIMULATED data, the Poisson prior fits, every (lambda, score) objective line and the
search results -- must come out of this repo's host driver + GPU objective in the same order."""
import gzip
import json
import math
import os

import pytest

from tests._transcript import parse_events

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, tol):
    if math.isinf(a) or math.isinf(b) or math.isnan(a) or math.isnan(b):
        return a == b or (math.isnan(a) and math.isnan(b))
    return abs(a - b) <= tol


def test_test3_transcript_search_genfamily_lhtest(tmp_path):
    from cafe_amd.shell import CafeShell
    gold = json.load(gzip.open(os.path.join(GOLD, "test3_transcript.json.gz"), "rt"))["events"]
    log = str(tmp_path / "log.txt")
    os.makedirs(tmp_path / "rndtree")
    sh = CafeShell(0, log)
    ltree = "(((2,2)1,(1,1)1)1,1)"
    for line in ["seed 10",
                 "load -i %s -p 0.01" % os.path.join(GOLD, "example_data.tab"),
                 "tree (((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)",
                 "lambda -s -t " + ltree,
                 # v4.1 (the transcript) printed the score of a plain `lambda -l`; v4.2.1 wants -score for it
                 "lambda -l 0.0017 -score",
                 "rootdist -i %s" % os.path.join(GOLD, "fly.table"),
                 "genfamily %s -t 10" % (tmp_path / "rndtree" / "rnd"),
                 "lhtest -d %s -l 0.0017 -t %s -o %s" % (tmp_path / "rndtree", ltree, tmp_path / "lh2.out")]:
        sh.dispatch(line)
    sh.close()
    mine = parse_events(open(log).read())
    kinds = lambda ev: [e[0] for e in ev]  # noqa: E731
    assert kinds(mine) == kinds(gold), "event sequence differs (first mismatch at %d)" % next(
        (i for i, (a, b) in enumerate(zip(kinds(mine), kinds(gold))) if a != b), min(len(mine), len(gold)))
    n_eval = 0
    for i, (a, b) in enumerate(zip(mine, gold)):
        if a[0] in ("families", "root_range", "family_range"):
            assert a == b, (i, a, b)           # sizes and ranges of the simulated tables: exact
        elif a[0] == "poisson":
            assert _close(a[1], b[1], 1.5e-6) and _close(a[2], b[2], 2e-6) and a[3] == b[3], (i, a, b)
        elif a[0] == "eval":
            assert len(a[1]) == len(b[1]) and all(_close(x, y, 6e-15) for x, y in zip(a[1], b[1])), (i, a, b)
            assert _close(a[2], b[2], 2e-6 + 2e-10 * abs(b[2])), (i, a, b)
            n_eval += 1
        elif a[0] == "result":
            assert a[1] == b[1], (i, a, b)
            assert all(_close(x, y, 6e-15) for x, y in zip(a[2], b[2])), (i, a, b)
            assert _close(a[3], b[3], 2e-6 + 2e-10 * abs(b[3])), (i, a, b)
    assert n_eval > 2000


def test_test4_transcript_error_model_two_classes(tmp_path):
    # tests/integration/test4.sh: seed 10; tree (12 taxa); load; errormodel -all -model errormodel.txt;
    # lambda -l 0.01 0.005 -t <2 classes> -score.  The transcript pins the table (12,653 families, ranges
    # 1~112 / 0~140) and the Poisson prior fit (0.761427 / 237754.098757 / 32 iterations); the score itself is
    # not printed there, so it is checked against the oracle (error model attached to every leaf).
    import shutil
    import numpy as np
    from cafe_amd.shell import CafeShell
    from tests import _orc as O
    g = json.load(open(os.path.join(GOLD, "transcripts.json")))["test4"]
    fam = str(tmp_path / "test4_families.txt")
    with gzip.open(os.path.join(GOLD, "test4_families.txt.gz"), "rb") as f, open(fam, "wb") as o:
        shutil.copyfileobj(f, o)
    log = str(tmp_path / "log.txt")
    sh = CafeShell(0, log)
    for line in ["seed 10", "tree " + g["newick"], "load -i " + fam,
                 "errormodel -all -model " + os.path.join(GOLD, "errormodel_test4.txt"),
                 "lambda -l 0.01 0.005 -t %s -score" % g["lambda_tree"]]:
        sh.dispatch(line)
    score, pl = sh.score, sh.poisson_lambda
    sh.close()
    ev = parse_events(open(log).read())
    assert ["families", g["n_families"]] in ev
    assert ["root_range"] + g["root_range"] in ev and ["family_range"] + g["family_range"] in ev
    po = [e for e in ev if e[0] == "poisson"][0]
    assert po[1] == pytest.approx(g["poisson_lambda"], abs=1.5e-6)
    assert po[2] == pytest.approx(g["poisson_score"], abs=2e-6) and po[3] == g["poisson_iters"]
    # oracle: same table, same error model, per-class lambdas
    sp, ids, counts = O.load_family_table(fam)
    t = O.PyTree(g["newick"])
    counts = O.reorder_to_tree(sp, counts, t)
    rng = O.range_from_max(int(counts.max()))
    assert (rng.root_min, rng.root_max, rng.min, rng.max) == (1, 112, 0, 140)
    # class of every node from the lambda tree (same topology, labels 1/2)
    cls = _classes_in_nlist_order(g["lambda_tree"])
    lam = np.where(np.array(cls) == 2, g["lambdas"][1], g["lambdas"][0])
    E, mfs = O.load_error_model(os.path.join(GOLD, "errormodel_test4.txt"), rng.max)
    prior = O.prior_poisson(1000, rng.root_min, pl)
    so, fz, *_ = O.eval_posterior(t, counts, rng, lam, np.full(t.n_nodes, -1.0), prior, errormatrix=E, err_mfs=mfs,
                                  leaf_has_err=np.ones(t.n_nodes, np.uint8),   # indexed by node id
                                  nthreads=8)
    assert fz < 0
    assert score == pytest.approx(-so, rel=1e-11)


def _classes_in_nlist_order(text):
    """Lambda-tree labels in the reference's in-order node numbering (leaf, internal, leaf, ...)."""
    pos = 0

    def parse():
        nonlocal pos
        if text[pos] == "(":
            pos += 1
            left = parse()
            assert text[pos] == ","
            pos += 1
            right = parse()
            assert text[pos] == ")"
            pos += 1
            start = pos
            while pos < len(text) and text[pos].isdigit():
                pos += 1
            label = int(text[start:pos]) if pos > start else 1
            return left + [label] + right
        start = pos
        while pos < len(text) and text[pos].isdigit():
            pos += 1
        return [int(text[start:pos])]

    return parse()
