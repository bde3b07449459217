"""Wall-clock of a BASELINE configs[4]-style pipeline on one GPU through the host driver: synthetic 32-taxon
table, error model on every leaf, lambda -s, report (Monte-Carlo null R x 1000 + Viterbi + p-values).
Usage: CAFEHOST_TIMING=1 python tools/cfg5_pipeline_time.py [families]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cafe_amd import synth
from cafe_amd.shell import CafeShell
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tree, counts, cfg = synth.make_config("cfg3", F=F)
names = ["t%d" % i for i in range(counts.shape[1])]
import re
# leaf names from the newick in order
leaf = re.findall(r"[(,]([A-Za-z0-9_]+):", cfg["newick"])
path = "/tmp/cfg5.tab"
with open(path, "w") as f:
    f.write("Desc\tFamily ID\t" + "\t".join(leaf) + "\n")
    for i, row in enumerate(counts):
        f.write("(null)\t%d\t%s\n" % (i, "\t".join(map(str, row))))
m = int(counts.max())
em = "/tmp/cfg5_err.txt"
with open(em, "w") as f:
    f.write("maxcnt:%d\ncntdiff -2 -1 0 1 2\n" % (m + max(50, m // 5)))
    f.write("0 0.00 0.00 0.94 0.05 0.01\n1 0.00 0.02 0.93 0.04 0.01\n")
    for j in range(2, m + max(50, m // 5) + 1):
        f.write("%d 0.01 0.02 0.93 0.03 0.01\n" % j)
sh = CafeShell(0, "/tmp/cfg5.log")
def t(cmd):
    t0 = time.perf_counter(); sh.dispatch(cmd); dt = time.perf_counter() - t0
    print("%-60s %8.3f s" % (cmd[:60], dt), flush=True)
t("seed 10")
t("load -i %s -t 1" % path)
t("tree " + cfg["newick"])
t("errormodel -model %s -all" % em)
t("lambda -s")
print("evaluations", sh.evaluations, "lambda", sh.params, "score", sh.score)
t("report /tmp/cfg5_report")
sh.close()
