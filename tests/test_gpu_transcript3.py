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
