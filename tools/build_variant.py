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
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libcafehip.so")
    cmd = [B.hipcc()] + B.FLAGS + flags + ["-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES]
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    print(out)


if __name__ == "__main__":
    main()
