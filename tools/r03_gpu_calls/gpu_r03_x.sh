#!/bin/bash
# round 3, call x: batched candidate evaluation forced on at BASELINE sizes
mkdir -p gpurun_out/r03x; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03x
(timeout 600 python tools/speculation_cfg.py cfg2 > $O/spec_cfg2.log 2>&1); grep "speculate=" $O/spec_cfg2.log | cut -c1-260
(timeout 600 python tools/speculation_cfg.py cfg2 3000 > $O/spec_cfg2_3k.log 2>&1); grep "speculate=" $O/spec_cfg2_3k.log | cut -c1-260
(timeout 900 python tools/speculation_cfg.py cfg3 > $O/spec_cfg3.log 2>&1); grep "speculate=" $O/spec_cfg3.log | cut -c1-260
