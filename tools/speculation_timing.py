#!/usr/bin/env python
"""Wall-clock of searches on a table that fills a fraction of the chip, sequential vs batched candidate
evaluation (cafehip_eval_posterior_multi): the shipped example (59 families, 5 taxa) -- `lambda -s`, `lambdamu -s`,
a two-class `lambda -s -t`, a `lambda -r` grid -- and lhtest over ten simulated example-sized tables.
Usage: python tools/speculation_timing.py"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
NEWICK = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"


def run(lines, speculate, reps=5):
    from cafe_amd.shell import CafeShell
    os.environ["CAFEHOST_SPECULATE"] = "1" if speculate else "0"
    best, res = 1e9, None
    for _ in range(reps):
        sh = CafeShell(0, os.devnull)
        for l in lines[:-1]:
            sh.dispatch(l)
        t0 = time.perf_counter()
        sh.dispatch(lines[-1])
        best = min(best, time.perf_counter() - t0)
        res = (list(sh.params), sh.score, sh.iterations, sh.evaluations, sh.speculation_stats())
        sh.close()
    return best, res


def main():
    import torch
    torch.cuda.init()
    base = ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + NEWICK]
    with tempfile.TemporaryDirectory() as d:
        for cmd in ("lambda -s", "lambdamu -s", "lambda -s -t (((1,1)1,(2,2)2)2,2)",
                    "lambda -r 0.0005:0.00025:0.02 -o %s/grid.txt" % d):
            t0, r0 = run(base + [cmd], False)
            t1, r1 = run(base + [cmd], True)
            same = r0[:4] == r1[:4]
            print("%-48s sequential %7.2f ms  batched %7.2f ms  x%.2f  evaluations %d  passes/points/hits %s  identical %s"
                  % (cmd.split(" -o")[0], 1e3 * t0, 1e3 * t1, t0 / t1, r0[3], r1[4], same), flush=True)
        # lhtest: ten simulated tables of the example's size
        sim = os.path.join(d, "sim")
        os.makedirs(sim)
        prep = base + ["lambda -s", "genfamily %s/rnd -t 10" % sim]
        lh = "lhtest -d %s -t (((1,1)1,(2,2)2)2,2) -l 0.0107527 -o %s/lh.out" % (sim, d)
        t0, r0 = run(prep + [lh], False, reps=2)
        out0 = open(os.path.join(d, "lh.out")).read()
        t1, r1 = run(prep + [lh], True, reps=2)
        out1 = open(os.path.join(d, "lh.out")).read()
        print("%-48s sequential %7.2f ms  batched %7.2f ms  x%.2f  identical output %s"
              % ("lhtest (10 tables x 2 searches)", 1e3 * t0, 1e3 * t1, t0 / t1, out0 == out1), flush=True)


if __name__ == "__main__":
    main()
