"""Where the one-time set-up of a table goes: CAFEHIP_SETUP_LOG=1 laps of cafehip_set_families on the
first and second load of the configs[1] / configs[3]-shard tables in one context, and on a second context of the process.
    CAFEHIP_SETUP_LOG=1 python tools/setup_laps.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cafe_amd
from cafe_amd import synth
from cafe_amd import tree as ctree

for cfg_name, F in (("cfg2", 10000), ("cfg4", 62500)):
    cfg = dict(synth.CONFIGS[cfg_name])
    tree = ctree.CafeTree(synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"])))
    counts = synth.simulate_families(tree, F, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    rng = cafe_amd.init_family_size(cfg["m"])
    for ctx_no in (1, 2):
        t0 = time.perf_counter()
        eng = cafe_amd.Engine(0)
        t1 = time.perf_counter()
        tree.apply(eng)
        t2 = time.perf_counter()
        sys.stderr.write("== %s context %d: create %.2f ms, set_tree %.2f ms\n" % (cfg_name, ctx_no, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        for load in (1, 2):
            t0 = time.perf_counter()
            eng.set_families(counts, rng)
            sys.stderr.write("== %s context %d load %d: set_families %.2f ms %s\n" % (cfg_name, ctx_no, load, 1e3 * (time.perf_counter() - t0), eng.last_setup_ms()))
        eng.close()
