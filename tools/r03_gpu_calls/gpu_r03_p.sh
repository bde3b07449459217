#!/bin/bash
# round 3, call p: tiles of a batch launch dealt in interleaved bands (co-resident workgroups of mixed extents)
mkdir -p gpurun_out/r03p; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03p
(timeout 1500 python tools/mcnull_one.py 5 "batch_mix=4" "batch_mix=16" "batch_mix=61" "batch_mix=251" "batch_lockstep=0;batch_mix=0" "batch_mix=4" "batch_mix=16" "batch_mix=61" "batch_mix=251" "batch_mix=1021" "batch_lockstep=1;batch_lockstep_slack=50;batch_mix=16" "batch_mix=61" "batch_lockstep_slack=0;batch_mix=0" 16:1,4,2,4 "batch_mix=16" "batch_mix=61" "batch_lockstep=0" "batch_mix=16" "batch_mix=251" > $O/mcnull_mix.log 2>&1); grep "^mcnull" $O/mcnull_mix.log | cut -c1-150
