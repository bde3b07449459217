#!/usr/bin/env python
"""A/B of library variants (tools/build_variant.py): `python tools/ab_variants.py main d0 d2 -- cfg2:10000 cfg3:100000`."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    cut = args.index("--")
    variants, specs = args[:cut], args[cut + 1:]
    for v in variants:
        env = dict(os.environ)
        if v != "main":
            env["CAFEHIP_LIB"] = os.path.join(ROOT, "tools", "_variants", v, "libcafehip.so")
        else:
            env.pop("CAFEHIP_LIB", None)
        print("=== variant %s" % v, flush=True)
        subprocess.call([sys.executable, os.path.join(ROOT, "tools", "ab_one.py")] + specs, env=env, cwd=ROOT)


if __name__ == "__main__":
    main()
