#!/usr/bin/env python
"""`lambda -s` on a BASELINE table (default cfg2: 10 k families, one workgroup per CU) with the candidates of a
Nelder-Mead iteration evaluated one by one or in one batched pass (host option speculate=0|1).
Usage: python tools/speculation_cfg.py [cfg2|cfg3|...] [families]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else None
    import torch
    torch.cuda.init()
    from cafe_amd import synth
    from cafe_amd.shell import CafeShell
    tree, counts, cfg = synth.make_config(name, F=F)
    has_mu = cfg["mu"] >= 0
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "families.tab")
        with open(path, "w") as f:
            f.write("Desc\tFamily ID\t" + "\t".join(tree.leaf_names) + "\n")
            for i, row in enumerate(counts):
                f.write("NA\tF%06d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")
        cmd = "lambdamu -s" if has_mu else ("lambda -s -t " + synth.clade_classes(tree, cfg["n_classes"])[1] if cfg.get("n_classes") else "lambda -s")
        for rep in range(2):
            for spec in ("0", "1"):
                sh = CafeShell(0, os.path.join(d, "log.txt"))
                sh.set_option("speculate", spec)
                sh.dispatch("seed 10")
                sh.dispatch("tree " + cfg["newick"])
                sh.dispatch("load -i " + path)
                t0 = time.perf_counter()
                sh.dispatch(cmd)
                wall = time.perf_counter() - t0
                print("%s %s speculate=%s wall %.2f ms  search %.2f ms  iterations %d evaluations %d  passes/points/hits %s  fitted %s score %.6f"
                      % (name, cmd[:12], spec, 1e3 * wall, 1e3 * sh.search_seconds, sh.iterations, sh.evaluations, sh.speculation_stats(),
                         ["%.10g" % x for x in sh.params], sh.score), flush=True)
                sh.close()


if __name__ == "__main__":
    main()
