#!/usr/bin/env python
"""Per-evaluation timeline from a rocprofv3 --kernel-trace rocpd database when evaluations may start at the pruning
(matrices built ahead of time, round 5): an evaluation ENDS with a score kernel (k3_score / k3_score_x); the launches
between two score kernels on the same queue are its chain; matrix builds that ran on another queue (the speculation
stream) are listed with their position relative to the chain they overlapped.

    chain_timeline.py <results.db> [n_last_evaluations=200]

Prints the most common chain shape among the last evaluations: start / end / duration of every launch relative to the
chain's first start, gaps, the turn-around (end of the score kernel -> first start of the next chain), and where the builds
of the other queue fell."""
import sqlite3
import sys


def short(name):
    for key in ("k1_build_matrices_rb", "k1_build_matrices", "k1e_fold_error", "k2c_gemm", "k2c_nodes", "k2_prune_mfma4", "k2_prune_mfma",
                "k2_prune_v1", "k3_score_x", "k3_score", "k3_cluster_score", "k_fetch_small", "k_x_collect", "ncclDevKernel", "rccl"):
        if key in name:
            return key
    return name[:32]


def columns(db, table):
    return [r[1] for r in db.execute("pragma table_info(%s)" % table)]


def main():
    db = sqlite3.connect(sys.argv[1])
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cols = columns(db, "kernels")
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")
    rows = [(short(r[0]), r[1], r[2], r[3] if qcol else 0) for r in db.execute(sel)]
    # the main queue is the one the score kernels run on
    main_q = None
    for r in rows:
        if r[0].startswith("k3_score"):
            main_q = r[3]
    chain_rows = [r for r in rows if r[3] == main_q]
    other_rows = [r for r in rows if r[3] != main_q]
    evals, cur = [], []
    for r in chain_rows:
        cur.append(r)
        if r[0].startswith("k3_score"):
            evals.append(cur)
            cur = []
    tail = evals[-n_last:]
    shapes = {}
    for ev in tail:
        shapes.setdefault(tuple(r[0] for r in ev), []).append(ev)
    print("evaluations in the trace: %d (queue column: %s; %d launches on other queues)" % (len(evals), qcol, len(other_rows)))
    for shape, group in sorted(shapes.items(), key=lambda kv: -len(kv[1]))[:3]:
        n = len(group)
        print("\nchain shape (%d of the last %d): %s" % (n, len(tail), " -> ".join(shape)))
        print("%-24s %10s %10s %10s %12s" % ("launch", "start_us", "end_us", "dur_us", "gap_before_us"))
        tot = 0.0
        for j, name in enumerate(shape):
            st = sum(ev[j][1] - ev[0][1] for ev in group) / n / 1e3
            en = sum(ev[j][2] - ev[0][1] for ev in group) / n / 1e3
            gap = 0.0 if j == 0 else sum(ev[j][1] - ev[j - 1][2] for ev in group) / n / 1e3
            tot += en - st
            print("%-24s %10.2f %10.2f %10.2f %12.2f" % (name, st, en, en - st, gap))
        span = sum(ev[-1][2] - ev[0][1] for ev in group) / n / 1e3
        idx = {id(ev): k for k, ev in enumerate(evals)}
        turns, periods = [], []
        for ev in group:
            k = idx[id(ev)]
            if k + 1 < len(evals):
                turns.append((evals[k + 1][0][1] - ev[-1][2]) / 1e3)
                periods.append((evals[k + 1][0][1] - ev[0][1]) / 1e3)
        turns.sort()
        periods.sort()
        print("GPU span first start -> last end: %.2f us (kernels %.2f us, gaps %.2f us)" % (span, tot, span - tot))
        if turns:
            print("turn-around score kernel end -> next chain's first start: median %.2f us, p10 %.2f, p90 %.2f (n=%d); period median %.2f us"
                  % (turns[len(turns) // 2], turns[len(turns) // 10], turns[9 * len(turns) // 10], len(turns), periods[len(periods) // 2]))
        # builds on the other queue that started inside these chains
        if other_rows:
            offs, durs, per_chain = [], [], []
            k = 0
            others = sorted(other_rows, key=lambda r: r[1])
            for ev in group:
                a, b = ev[0][1], ev[-1][2]
                inside = [r for r in others if a <= r[1] <= b]
                per_chain.append(len(inside))
                for r in inside:
                    offs.append((r[1] - a) / 1e3)
                    durs.append((r[2] - r[1]) / 1e3)
            if offs:
                offs.sort()
                durs.sort()
                print("launches of the other queue that started inside these chains: %.2f per chain; start offset median %.1f us (p10 %.1f, p90 %.1f); "
                      "duration median %.1f us (p90 %.1f)" % (sum(per_chain) / len(per_chain), offs[len(offs) // 2], offs[len(offs) // 10],
                                                                offs[9 * len(offs) // 10], durs[len(durs) // 2], durs[9 * len(durs) // 10]))


if __name__ == "__main__":
    main()
