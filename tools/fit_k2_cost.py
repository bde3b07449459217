#!/usr/bin/env python
"""Re-fit the constants of the K2 wave-grid cost model (k2_cost / choose_mfma*_cfg in cafehip.hip) on sweep data:
files written by tools/sweep_k2.py and tools/sweep_k2_4x4.py (one per workload).  Prints the regret of the
current constants and of the best constants found by a random search."""
import glob
import os
import random
import re
import sys

N_CU = 256
SHAPES = {"cfg2": 10, "cfg3": 16, "cfg4": 10}   # row tiles of C


def load(dirname):
    data = {}
    for f in sorted(glob.glob(os.path.join(dirname, "s*_cfg*_*.txt"))):
        m = re.match(r"s(4|16)_(cfg\d)_(\d+)\.txt", os.path.basename(f))
        kind, cfg, n = m.group(1), m.group(2), int(m.group(3))
        for line in open(f):
            t = re.search(r"k2 ([0-9.]+)", line)
            c = re.search(r"cfg\(nftw,nrtw,wf,wr\)=(\d+),(\d+),(\d+),(\d+)", line)
            is4 = "mfma4x4" in line
            if not t or not c:
                continue
            key = (cfg, n)
            cfgt = (4 if is4 else 16,) + tuple(int(x) for x in c.groups())
            ms = float(t.group(1))
            d = data.setdefault(key, {})
            d[cfgt] = min(ms, d.get(cfgt, 1e9))
    return data


def k2_cost(P, n_items, nf, groups, wf, wr, RTc):
    W = wf * wr
    n_wg = (n_items + nf - 1) // nf
    wg_on_cu = (n_wg + N_CU - 1) // N_CU
    per_wg, active = 0, 0
    for w in range(W):
        wrow = w // wf
        act = RTc // wr + (1 if wrow < RTc % wr else 0)
        per_wg += act * groups
        active += act > 0
    simds = min(4, max(1, wg_on_cu * active))
    maxload = wg_on_cu * per_wg / simds
    cost = maxload * P["wpen"][W]
    cost *= 1.0 + P["restream"] * (n_wg * wf) / N_CU
    hi_t = RTc // wr + (1 if RTc % wr else 0)
    cost *= 1.0 + P["imb"] * (hi_t / (RTc / wr) - 1.0)
    return cost


def model(P, kind, nft, nrt, wf, wr, n_items, RTc):
    if kind == 16:
        return k2_cost(P, n_items, 16 * nft * wf, 4 * nft, wf, wr, RTc)
    return P["f4"] * (1.0 + P["gpen"] / (nft * nft)) * k2_cost(P, n_items, 4 * nft * wf, nft, wf, wr, RTc)


def regret(P, data, verbose=False):
    tot = 0.0
    for (cfg, n), d in sorted(data.items()):
        RTc = SHAPES[cfg]
        best_ms = min(d.values())
        pick = min(d, key=lambda k: model(P, k[0], k[1], k[2], k[3], k[4], n, RTc))
        r = d[pick] / best_ms - 1.0
        tot += r
        if verbose:
            bk = min(d, key=d.get)
            print("%s %7d: picks %s %.3f ms, best %s %.3f ms  regret %.1f %%" % (cfg, n, pick, d[pick], bk, best_ms, 100 * r))
    return tot / len(data)


def main():
    data = load(sys.argv[1] if len(sys.argv) > 1 else "profiles/r01_k2_sweeps")
    P0 = {"wpen": [0, 1.1, 1.0, 1.0, 1.0, 0.975, 0.95, 0.925, 0.9], "restream": 0.001, "imb": 0.2, "f4": 1.07, "gpen": 0.5}  # as in cafehip.hip
    print("current constants: mean regret %.2f %%" % (100 * regret(P0, data, True)))
    rnd = random.Random(1)
    best, bestP = regret(P0, data), P0
    for it in range(20000):
        P = {"wpen": list(bestP["wpen"]), "restream": bestP["restream"], "imb": bestP["imb"], "f4": bestP["f4"], "gpen": bestP["gpen"]}
        k = rnd.choice(["wpen", "wpen", "restream", "imb", "f4", "gpen"])
        if k == "wpen":
            i = rnd.choice([1, 2, 4, 8])
            P["wpen"][i] = max(0.7, P["wpen"][i] * (1 + rnd.uniform(-0.08, 0.08)))
        elif k == "restream":
            P[k] = max(0.0, P[k] + rnd.uniform(-0.001, 0.001))
        else:
            P[k] = max(0.0, P[k] * (1 + rnd.uniform(-0.1, 0.1)))
        r = regret(P, data)
        if r < best - 1e-9:
            best, bestP = r, P
    print("fitted constants: mean regret %.2f %%" % (100 * regret(bestP, data, True)))
    print(bestP)


if __name__ == "__main__":
    main()
