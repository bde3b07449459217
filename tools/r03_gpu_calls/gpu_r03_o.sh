#!/bin/bash
# round 3, call o: per-phase stamps of the Monte-Carlo-null launch by trimmed extent
mkdir -p gpurun_out/r03o; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03o
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 900 python tools/k2_stamps_null.py 1000 > $O/stamps_null.log 2>&1); tail -14 $O/stamps_null.log | cut -c1-250
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 900 python tools/k2_stamps_null.py 1000 K2CFG=1,4,2,4 > $O/stamps_null_1424.log 2>&1); tail -10 $O/stamps_null_1424.log | cut -c1-250
