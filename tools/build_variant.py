#!/usr/bin/env python
"""Build a variant of libcafehip.so with extra -D flags for A/B runs and debug timelines:

    python tools/build_variant.py stamps -DCAFE_K2_STAMPS
    python tools/build_variant.py d4 -DCAFE_K2_DEPTH4=4
    CAFEHIP_LIB=tools/_variants/d0/libcafehip.so python bench.py ...

Outputs go to tools/_variants/<name>/ (git-ignored, travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cafe_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "tools", "_variants", name)
    obj_dir = os.path.join(out_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(out_dir, "libcafehip.so")
    jobs, objs = [], []
    for src in B.SOURCES:   # the translation units side by side, as cafe_amd/build.py compiles them
        obj = os.path.join(obj_dir, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        jobs.append([B.hipcc()] + B.CFLAGS + B.UNIT_FLAGS.get(src, []) + flags + ["-c", "-o", obj, os.path.join(B.CSRC, src)])
    print(" ".join(jobs[0]), "... (%d units)" % len(jobs), flush=True)
    procs = [subprocess.Popen(j) for j in jobs]
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("hipcc failed")
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    print(out)


if __name__ == "__main__":
    main()
