"""Search wall-clock with and without the matrices of the optimiser's next points built ahead of time (host option
`lookahead`), same process, alternating, several repetitions (round 5).

    python tools/lookahead_ab.py [cfg2|cfg3|cfg4|test1|example ...] [--reps N]

Prints, per table: search seconds (the Nelder-Mead loop only, `cafehost_search_seconds`) of every repetition, evaluations,
how many were served from matrices built ahead, and whether the two runs asked for the same points and got the same values."""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def table(name, d):
    """-> (lines of the cafe script up to the search, the search command)"""
    from cafe_amd import synth
    import cafe_amd
    if name == "example":
        return ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"),
                "tree (((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"], "lambda -s"
    if name == "test1":
        TR = json.load(open(os.path.join(GOLD, "transcripts.json")))
        path = os.path.join(d, "test1_families.txt")
        with open(path, "wb") as f:
            f.write(gzip.open(os.path.join(GOLD, "test1_families.txt.gz")).read())
        return ["seed 10", "tree " + TR["test1"]["newick"], "load -i %s -max_size 20" % path], "lambda -s"
    tree, counts, cfg = synth.make_config(name)
    if name == "cfg4":
        counts = counts[:62464]
    path = os.path.join(d, name + ".tab")
    with open(path, "w") as f:
        f.write("Desc\tFamily ID\t" + "\t".join(tree.leaf_names) + "\n")
        for i, row in enumerate(counts):
            f.write("NA\tF%06d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")
    lines = ["seed 10", "tree " + cfg["newick"], "load -i " + path]
    if cfg.get("error_model"):
        em = os.path.join(d, "errormodel.txt")
        synth.write_error_model_file(em, cafe_amd.init_family_size(cfg["m"]).max)
        lines.append("errormodel -model %s -all" % em)
    if cfg["mu"] >= 0:
        return lines, "lambdamu -s"
    if cfg.get("n_classes"):
        return lines, "lambda -s -t " + synth.clade_classes(tree, cfg["n_classes"])[1]
    return lines, "lambda -s"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tables", nargs="*", default=["cfg2", "test1"])
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--speculate", default=None, help="host option speculate for both runs (default: the library's choice)")
    ap.add_argument("--option", action="append", default=[], help="key=value handed to the session (e.g. matrix_cache=0: the look-ahead's host work without the builds)")
    a = ap.parse_args()
    from cafe_amd.shell import CafeShell
    for name in a.tables:
        with tempfile.TemporaryDirectory() as d:
            lines, command = table(name, d)
            sh = CafeShell(0, os.devnull)
            if a.speculate is not None:
                sh.set_option("speculate", a.speculate)
            for l in lines:
                sh.dispatch(l)
            for kv in a.option:
                k, v = kv.split("=", 1)
                sh.set_option(k, v)
            times = {0: [], 1: []}
            results = {}
            stats = None
            for rep in range(a.reps):
                for look in (0, 1):
                    sh.set_option("lookahead", look)
                    sh.dispatch("seed 10")
                    before = sh.lookahead_stats()
                    t0 = time.perf_counter()
                    sh.dispatch(command)
                    wall = time.perf_counter() - t0
                    times[look].append((sh.search_seconds, wall))
                    results[look] = (list(sh.params), sh.score, sh.iterations, sh.evaluations, sh.trace().tolist())
                    if look:
                        after = sh.lookahead_stats()
                        stats = {k: after[k] - before[k] for k in after}
            same = results[0] == results[1]
            ev = results[1][3]
            print("%-8s %-14s %4d evaluations, %4d served from matrices built ahead (%d sets built, %d announcements); same trajectory: %s"
                  % (name, command.split(" -t")[0], ev, stats["hits"], stats["built"], stats["announcements"], same))
            for look in (0, 1):
                print("    lookahead=%d  search ms: %s   (whole command ms: %s)" % (
                    look, " ".join("%.3f" % (1e3 * s) for s, w in times[look]), " ".join("%.2f" % (1e3 * w) for s, w in times[look])))
            b0 = min(s for s, w in times[0][1:] or times[0])
            b1 = min(s for s, w in times[1][1:] or times[1])
            print("    best after the first repetition: %.3f -> %.3f ms (%+.1f %%), per evaluation %.1f -> %.1f us"
                  % (1e3 * b0, 1e3 * b1, 100 * (b1 / b0 - 1), 1e6 * b0 / ev, 1e6 * b1 / ev))
            sh.close()


if __name__ == "__main__":
    main()
