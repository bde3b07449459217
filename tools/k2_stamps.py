#!/usr/bin/env python
"""Phase timeline of one K2 launch from the s_memtime stamps of a -DCAFE_K2_STAMPS build.

    python tools/build_variant.py stamps -DCAFE_K2_STAMPS
    CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so python tools/k2_stamps.py cfg2 [families] [K2CFG4=G,nrtw,wf,wr]

Prints, per step of the walk (averaged over workgroups; a step's time is taken on the workgroup's slowest wave):
the kind of step (LL = two gathered children: leaves or compressed subtrees, LI = one gathered + one internal child,
II = two internal children), shader cycles from
the end of the previous step to: gathers issued, factors done, read barrier passed, result visible."""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    F = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
    for arg in sys.argv[2:]:
        if "=" in arg:
            k, v = arg.split("=", 1)
            os.environ["CAFEHIP_" + k] = v
    os.environ.setdefault("CAFEHIP_K2TUNE", "0")
    path = os.path.join(tempfile.gettempdir(), "k2_stamps.bin")
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth, prior as cprior
    tree, counts, cfg = synth.make_config(name, F=F)
    rng = cafe_amd.init_family_size(cfg["m"])
    lam_p = cprior.poisson_lambda_mle(counts)
    prior = cprior.prior_rfsize_poisson(rng.root_min, lam_p)
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    nl = np.full(tree.n_nodes, cfg["lam"])
    nm = np.full(tree.n_nodes, cfg["mu"])
    for _ in range(5):
        eng.get_posterior(nl, nm, prior)          # warm
    os.environ["CAFEHIP_STAMPS_FILE"] = path
    eng.enable_timing(True)
    eng.get_posterior(nl, nm, prior)
    ms = eng.last_kernel_ms()
    print(eng.describe())
    print("kernel ms (with stamps): k1 %.3f k2 %.3f k3 %.3f" % tuple(ms))
    raw = open(path, "rb").read()
    hdr = struct.unpack("8q", raw[:64])
    grid, waves, slots, n_ops, nf, shape, wf, wr = hdr
    ops = np.frombuffer(raw[64:64 + 48 * n_ops], np.int32).reshape(n_ops, 12)
    st = np.frombuffer(raw[64 + 48 * n_ops:], np.uint64).reshape(grid, 8, slots)[:, :waves, :].astype(np.int64)
    last = 2 + 6 * n_ops
    print("grid %d, %d waves/workgroup, NF %d, shape %dx, Wf %d Wr %d, %d steps" % (grid, waves, nf, shape, wf, wr, n_ops))
    t0 = st[:, :, 0].min(axis=1)
    total = st[:, :, last].max(axis=1) - t0
    print("workgroup total cycles: mean %.0f  min %d  max %d   (%.1f us at 2.4 GHz)" %
          (total.mean(), total.min(), total.max(), total.mean() / 2400.0))
    pro = st[:, :, 1].max(axis=1) - t0
    print("prologue (counts, step list -> LDS): %.0f cycles" % pro.mean())
    prev_end = st[:, :, 1].max(axis=1)
    acc = {"LL": [0, 0.0], "LI": [0, 0.0], "II": [0, 0.0]}
    print("%4s %4s %5s %9s %9s %9s %9s %9s | %9s" % ("step", "kind", "root", "gathers", "factor1", "factor2", "barrier", "visible",
                                                      "wave-skew"))
    for oi in range(n_ops):
        kinds = ops[oi, 4:6]
        kind = {0: "LL", 1: "LI", 2: "II"}[int((kinds == 1).sum())]   # children of kind 0 (leaf) / 2 (compressed subtree) are gathers
        b = 2 + 6 * oi
        def rel(slot, red="max"):
            v = st[:, :, b + slot]
            v = np.where(v == 0, np.int64(0), v)
            m = v.max(axis=1) if red == "max" else v.min(axis=1)
            return np.where(m > 0, m - prev_end, 0)
        g = rel(0)
        f1 = rel(1)
        f2 = rel(2)
        bar = rel(3)
        vis = rel(4)
        # skew: difference between the slowest and the fastest wave reaching "factors done"
        fd = np.maximum(st[:, :, b + 1], st[:, :, b + 2])
        skew = (fd.max(axis=1) - fd.min(axis=1)).mean()
        print("%4d %4s %5d %9.0f %9.0f %9.0f %9.0f %9.0f | %9.0f" % (oi, kind, ops[oi, 1], g.mean(), f1.mean(), f2.mean(),
                                                                   bar.mean(), vis.mean(), skew))
        acc[kind][0] += 1
        acc[kind][1] += vis.mean()
        prev_end = st[:, :, b + 4].max(axis=1)
    epi = st[:, :, last].max(axis=1) - prev_end
    print("epilogue (posterior): %.0f cycles" % epi.mean())
    if st[:, :, last + 1].max() > 0:
        e1 = st[:, :, last + 1].max(axis=1) - prev_end
        e2 = st[:, :, last + 2].max(axis=1) - prev_end
        e3 = st[:, :, last + 3].max(axis=1) - prev_end
        print("   prior in registers %.0f, families scanned %.0f, candidates evaluated %.0f, outputs written %.0f" %
              (e1.mean(), e2.mean(), e3.mean(), epi.mean()))
    for k, (n, t) in acc.items():
        if n:
            print("%s steps: %d, mean %.0f cycles each, %.0f total (%.1f %% of the workgroup time)" %
                  (k, n, t / n, t, 100.0 * t / total.mean()))
    eng.close()


if __name__ == "__main__":
    main()
