#!/usr/bin/env python
"""Factor-table launches (and the whole evaluation) under several option sets, same process, alternating (round 6: k2c_gemm
against k2c_nodes):

    python tools/k2c_ab.py cfg2:10000 cfg3:100000 test1 -- k2c_gemm=0 k2c_gemm=1,k2c_nst=4 k2c_gemm=1,k2c_nst=2,k2c_pair=1

Per workload and option set: tables ms (HIP events around the k2c launches), walk ms, step ms (host clock over 20
evaluations), the score's bits (every set must print the same), the plan.  Tables are cached under /tmp."""
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")


def workload(spec):
    import cafe_amd
    from cafe_amd import prior as cprior, synth
    if spec == "test1":
        from cafe_amd.tree import CafeTree
        TR = json.load(open(os.path.join(GOLD, "transcripts.json")))
        tree = CafeTree(TR["test1"]["newick"])
        rows = []
        with gzip.open(os.path.join(GOLD, "test1_families.txt.gz"), "rt") as f:
            header = f.readline().rstrip("\n").split("\t")
            names = [h.lower() for h in header[2:]]
            col = [names.index(n.lower()) for n in tree.leaf_names]
            for line in f:
                p = line.rstrip("\n").split("\t")
                r = [int(p[2 + c]) for c in col]
                if max(r) <= 20:
                    rows.append(r)
        counts = np.array(rows, np.int32)
        cfg = {"lam": 0.008, "mu": -1.0}
        m = int(counts.max())
        nl = np.full(tree.n_nodes, 0.008)
        nm = np.full(tree.n_nodes, -1.0)
    else:
        from ab_one import table
        name, F = spec.split(":")
        tree, counts, cfg = table(name, int(F))
        m = cfg["m"]
        nl, nm = synth.node_rates(tree, cfg)
    rng = cafe_amd.init_family_size(m)
    prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
    return tree, counts, cfg, rng, prior, nl, nm


def main():
    args = sys.argv[1:]
    cut = args.index("--")
    specs, sets = args[:cut], args[cut + 1:]
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth
    for spec in specs:
        tree, counts, cfg, rng, prior, nl, nm = workload(spec)
        engines = []
        for st in sets:
            eng = cafe_amd.Engine(0)
            for kv in st.split(","):
                if kv and kv != "default":
                    k, v = kv.split("=")
                    eng.set_option(k, v)
            tree.apply(eng)
            eng.set_families(counts, rng)
            if cfg.get("error_model"):
                eng.set_error_model(synth.banded_error_matrix(rng.max))
            n = 0
            t0 = time.perf_counter()
            while n < 40 or (time.perf_counter() - t0 < 0.3 and n < 300):   # wave grid settled, clocks up
                eng.get_posterior(nl, nm, prior)
                n += 1
            engines.append(eng)
        res = {st: [] for st in sets}
        for rep in range(3):
            for st, eng in zip(sets, engines):
                eng.enable_timing(True)
                ks = []
                for _ in range(12):
                    score, fz = eng.get_posterior(nl, nm, prior)
                    ks.append(eng.last_kernel_ms() + [eng.last_tables_ms()])
                eng.enable_timing(False)
                t0 = time.perf_counter()
                for _ in range(20):
                    eng.get_posterior(nl, nm, prior)
                step = (time.perf_counter() - t0) / 20 * 1e3
                ks = np.array(ks)
                res[st].append((np.median(ks[:, 3]), np.median(ks[:, 1] - ks[:, 3]), step, np.median(ks[:, 0]), np.median(ks[:, 2]), score))
        print("== %s (%d rows)" % (spec, len(counts)))
        for st, eng in zip(sets, engines):
            r = np.array([x[:5] for x in res[st]])
            print("  %-44s tables %s  (best %.4f)  walk %.4f  k1 %.4f  k3 %.4f  step %.4f ms  score %s" % (
                st, " ".join("%.4f" % x for x in r[:, 0]), r[:, 0].min(), r[:, 1].min(), r[:, 3].min(), r[:, 4].min(), r[:, 2].min(), float(res[st][-1][5]).hex()), flush=True)
            eng.close()


if __name__ == "__main__":
    main()
