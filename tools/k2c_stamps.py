#!/usr/bin/env python
"""Phase timeline of the k2c_nodes launches (factor tables of compressed subtrees) of ONE evaluation, from the s_memtime
stamps of a -DCAFE_K2_STAMPS build:

    python tools/build_variant.py stamps -DCAFE_K2_STAMPS
    CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so python tools/k2c_stamps.py cfg2 [families]

Per level: tiles, and the mean over tiles (slowest wave) of the cycles from the tile's start to: header loaded, matrix
indices loaded, columns gathered and written to LDS, barrier passed, product done, rows stored."""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else None
    path = os.path.join(tempfile.gettempdir(), "k2c_stamps.bin")
    for p in (path, path + ".k2c"):
        if os.path.exists(p):
            os.remove(p)
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth, prior as cprior
    tree, counts, cfg = synth.make_config(name, F=F)
    rng = cafe_amd.init_family_size(cfg["m"])
    prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    nl, nm = synth.node_rates(tree, cfg)
    for _ in range(40):
        eng.get_posterior(nl, nm, prior)
    os.environ["CAFEHIP_STAMPS_FILE"] = path
    eng.get_posterior(nl, nm, prior)
    del os.environ["CAFEHIP_STAMPS_FILE"]
    print(eng.describe())
    eng.close()
    raw = open(path + ".k2c", "rb").read()
    off, level = 0, 0
    names = ["header", "indices", "gathered", "barrier", "product", "stored"]
    print("%5s %6s %6s | %s | %9s" % ("level", "tiles", "waves", " ".join("%9s" % n for n in names), "mean tile"))
    while off < len(raw):
        grid, waves, slots, _ = struct.unpack("4q", raw[off:off + 32])
        off += 32
        n = grid * 16 * slots
        st = np.frombuffer(raw[off:off + 8 * n], np.uint64).reshape(grid, 16, slots)[:, :waves, :].astype(np.int64)
        off += 8 * n
        t0 = st[:, :, 0].min(axis=1)
        cols = [(st[:, :, k].max(axis=1) - t0).mean() for k in range(1, 7)]
        # (s_memtime counters of different XCDs are not synchronised: only differences inside a tile mean anything)
        print("%5d %6d %6d | %s | %9.0f" % (level, grid, waves, " ".join("%9.0f" % c for c in cols), cols[-1]))
        if os.environ.get("K2C_STAMPS_DETAIL"):
            # per wave, from the wave's OWN start: how long each stage takes once the wave runs, and how far apart the waves of a
            # tile start
            own = [(st[:, :, k] - st[:, :, 0]) for k in range(1, 7)]
            skew = (st[:, :, 0].max(axis=1) - t0)
            print("      launch skew inside a tile: mean %.0f  p50 %.0f  p90 %.0f  max %.0f" % (skew.mean(), np.percentile(skew, 50), np.percentile(skew, 90), skew.max()))
            print("      per wave from its own start (mean / p10 / p90): " + "  ".join(
                "%s %.0f/%.0f/%.0f" % (names[k], own[k].mean(), np.percentile(own[k], 10), np.percentile(own[k], 90)) for k in range(6)))
            seg = [own[0]] + [own[k] - own[k - 1] for k in range(1, 6)]
            print("      per wave, stage by stage (mean): " + "  ".join("%s %.0f" % (names[k], seg[k].mean()) for k in range(6)))
        level += 1


if __name__ == "__main__":
    main()
