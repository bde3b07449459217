#!/usr/bin/env python
"""Registers / scratch of every K2 instantiation from the device assembly:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/w16.s cafe_amd/csrc/k2_walk16.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/w4.s cafe_amd/csrc/k2_walk4.hip
    python tools/k2_regs.py /tmp/w16.s /tmp/w4.s
A line marked SPILL means scratch memory is used: drop that wave tile from the launcher's candidate set
(k2_fits4 / kMaxTiles16 in cafehip.hip)."""
import re
import sys

for f in sys.argv[1:]:
    name, out = None, []
    for line in open(f):
        m = re.match(r'^(_ZN12_GLOBAL__N_1\d+k2_prune_(mfma4?)ILi(\d+)ELi(\d+)EEEvN\w*10K2MfmaArgsE):', line)
        if m:
            name = (m.group(2), int(m.group(3)), int(m.group(4)))
            v = sc = None
        if name:
            m2 = re.match(r'^; NumVgprs: (\d+)', line)
            if m2:
                v = int(m2.group(1))
            m3 = re.match(r'^; ScratchSize: (\d+)', line)
            if m3:
                sc = int(m3.group(1))
            m4 = re.match(r'^; Occupancy: (\d+)', line)
            if m4:
                out.append((name, v, sc, int(m4.group(1))))
                name = None
    for n, v, sc, occ in sorted(out):
        print("%s<%d,%d> vgpr %d scratch %d waves/SIMD %d%s" % (n[0], n[1], n[2], v, sc, occ, "  SPILL" if sc else ""))
