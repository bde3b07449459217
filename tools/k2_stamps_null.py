#!/usr/bin/env python
"""Where the Monte-Carlo-null launch (BASELINE configs[4]: 250 root sizes x `trials` simulated rows, batch mode) spends
its time, from the s_memtime stamps of a -DCAFE_K2_STAMPS build, by the trimmed extent of the workgroup's tile.

    python tools/build_variant.py stamps -DCAFE_K2_STAMPS
    CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so python tools/k2_stamps_null.py [trials] [K2CFG=a,b,wf,wr]

Per band of column limits: workgroups, mean cycles per workgroup, and the mean cycles per phase summed over the walk's
steps (gathers issued, first factor done, second factor done, read barrier passed, result visible), the prologue and
the root output."""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
    for arg in sys.argv[1:]:
        if "=" in arg:
            k, v = arg.split("=", 1)
            os.environ["CAFEHIP_" + k] = v
    path = os.path.join(tempfile.gettempdir(), "k2_stamps_null.bin")
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth
    from cafe_amd import prior as cprior
    tree, counts, cfg = synth.make_config("cfg5", F=20000)
    rng = cafe_amd.init_family_size(cfg["m"])
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    lam, mu = synth.node_rates(tree, cfg)
    prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
    for _ in range(45):
        eng.get_posterior(lam, mu, prior)
    eng.reset_birthdeath_cache(lam, mu)
    mats = {v: eng.get_matrix(v) for v in range(tree.n_nodes) if v != tree.root}
    rows, lo, cm = synth.simulate_null_rows(tree, mats, rng, trials, cfg["seed"] + 77)
    eng.enable_timing(True)
    for _ in range(2):
        eng.eval_root_likelihoods(rows, lo, lo, cm)
    os.environ["CAFEHIP_STAMPS_FILE"] = path
    eng.eval_root_likelihoods(rows, lo, lo, cm)
    print("launch ms (with stamps) %.3f  %s" % (eng.last_batch_ms(), eng.describe()))
    raw = open(path, "rb").read()
    grid, waves, slots, n_ops, nf, shape, wf, wr = struct.unpack("8q", raw[:64])
    ops = np.frombuffer(raw[64:64 + 48 * n_ops], np.int32).reshape(n_ops, 12)
    st = np.frombuffer(raw[64 + 48 * n_ops:], np.uint64).reshape(grid, 8, slots)[:, :waves, :].astype(np.int64)
    last = 2 + 6 * n_ops
    print("grid %d, %d waves/workgroup, NF %d, shape %dx, Wf %d Wr %d, %d steps" % (grid, waves, nf, shape, wf, wr, n_ops))
    cmx = np.zeros(grid, np.int64)
    pad = np.zeros(grid * nf, np.int64)
    pad[:len(cm)] = cm
    cmx = pad.reshape(grid, nf).max(axis=1)
    t0 = st[:, :, 0].min(axis=1)
    total = st[:, :, last].max(axis=1) - t0
    start = t0 - t0.min()
    print("launch span %.0f cycles (100 MHz memtime ticks if < 1e7: %.3f ms)" % ((st[:, :, last].max() - t0.min()), (st[:, :, last].max() - t0.min()) / 1e5))
    bands = [(0, 64), (64, 96), (96, 128), (128, 160), (160, 208), (208, 256), (256, 400)]
    phases = ["gathers", "factor1", "factor2", "barrier", "visible"]
    print("%10s %6s %9s %9s | %s | %9s" % ("col limit", "wgs", "cycles", "prologue", " ".join("%9s" % p for p in phases), "root out"))
    for lo_, hi_ in bands:
        m = (cmx >= lo_) & (cmx < hi_)
        if not m.any():
            continue
        s = st[m]
        prev_end = s[:, :, 1].max(axis=1)
        pro = (prev_end - t0[m]).mean()
        sums = np.zeros(5)
        for oi in range(n_ops):
            b = 2 + 6 * oi
            prevp = prev_end
            for ph in range(5):
                v = s[:, :, b + ph].max(axis=1)
                v = np.where(v > 0, v, prevp)
                sums[ph] += (v - prevp).mean()
                prevp = v
            prev_end = s[:, :, b + 4].max(axis=1)
        epi = (s[:, :, last].max(axis=1) - prev_end).mean()
        print("%4d..%-4d %6d %9.0f %9.0f | %s | %9.0f" % (lo_, hi_ - 1, m.sum(), total[m].mean(), pro, " ".join("%9.0f" % x for x in sums), epi))
    eng.close()


if __name__ == "__main__":
    main()
