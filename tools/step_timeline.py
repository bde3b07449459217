#!/usr/bin/env python
"""Per-evaluation timeline from a rocprofv3 --kernel-trace rocpd database: for the steady-state evaluations of a
bench run (K1 matrix build -> factor-table levels -> family walk -> K3 score), the start and end of every launch
relative to the K1 start, the gaps between launches, and the turn-around from the end of one evaluation's last
kernel to the start of the next evaluation's first (host: result pick-up, parameter staging, launch).

    step_timeline.py <results.db> [n_last_evaluations=200]"""
import sqlite3
import sys


def short(name):
    for key in ("k1_build_matrices_rb", "k1_build_matrices", "k1e_fold_error", "k2c_gemm", "k2c_nodes", "k2_prune_mfma4", "k2_prune_mfma",
                "k2_prune_v1", "k3_score", "k3_cluster_score", "k_fetch_small", "k_exchange", "ncclDevKernel", "rccl"):
        if key in name:
            return key
    return name[:32]


def main():
    db = sqlite3.connect(sys.argv[1])
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(short(n), s, e) for n, s, e in rows]
    # an evaluation starts at a K1 launch and runs to the launch before the next K1
    starts = [i for i, r in enumerate(rows) if r[0].startswith("k1_build")]
    evals = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
    # steady state: the most common launch sequence among the last n_last evaluations
    tail = evals[-n_last - 1:-1] if len(evals) > n_last + 1 else evals[:-1]
    shapes = {}
    for ev in tail:
        shapes.setdefault(tuple(r[0] for r in ev), []).append(ev)
    shape, group = max(shapes.items(), key=lambda kv: len(kv[1]))
    print("evaluations in the trace: %d; steady-state shape (%d of the last %d): %s" % (len(evals), len(group), len(tail), " -> ".join(shape)))
    n = len(group)
    print("%-24s %10s %10s %10s %12s" % ("launch", "start_us", "end_us", "dur_us", "gap_before_us"))
    tot_dur = 0.0
    for j, name in enumerate(shape):
        st = sum(ev[j][1] - ev[0][1] for ev in group) / n / 1e3
        en = sum(ev[j][2] - ev[0][1] for ev in group) / n / 1e3
        gap = 0.0 if j == 0 else sum(ev[j][1] - ev[j - 1][2] for ev in group) / n / 1e3
        tot_dur += en - st
        print("%-24s %10.2f %10.2f %10.2f %12.2f" % (name, st, en, en - st, gap))
    span = sum(ev[-1][2] - ev[0][1] for ev in group) / n / 1e3
    # turn-around: consecutive evaluations of the steady-state shape
    idx = {id(ev): k for k, ev in enumerate(evals)}
    turns = []
    for ev in group:
        k = idx[id(ev)]
        if k + 1 < len(evals) and tuple(r[0] for r in evals[k + 1]) == shape:
            turns.append((evals[k + 1][0][1] - ev[-1][2]) / 1e3)
    turns.sort()
    print("GPU span first start -> last end: %.2f us (kernels %.2f us, gaps between launches %.2f us)" % (span, tot_dur, span - tot_dur))
    if turns:
        print("turn-around last end -> next evaluation's first start (host): median %.2f us, p10 %.2f, p90 %.2f  (n=%d)"
              % (turns[len(turns) // 2], turns[len(turns) // 10], turns[9 * len(turns) // 10], len(turns)))
        print("=> evaluation period ~ %.2f us" % (span + turns[len(turns) // 2]))


if __name__ == "__main__":
    main()
