#!/bin/bash
# round 3, call ab: randomised soak of the C ABI against the oracle
mkdir -p gpurun_out/r03ab; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ab
(timeout 2300 python tests/soak_fuzz.py ${1:-420} ${2:-1} > $O/soak.log 2>&1; echo "rc=$?" >> $O/soak.log); grep -c "^ok" $O/soak.log; grep "FAIL\|Error\|error\|Traceback" $O/soak.log | head -20; tail -3 $O/soak.log | cut -c1-300
