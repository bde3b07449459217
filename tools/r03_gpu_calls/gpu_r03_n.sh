#!/bin/bash
# round 3, call n: wave grids of the Monte-Carlo-null launch now that a trimmed tile issues only its own row tiles
mkdir -p gpurun_out/r03n; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03n
(timeout 1500 python tools/mcnull_one.py 5 16:1,4,1,4 16:1,4,2,4 16:2,4,1,4 16:1,2,1,8 16:2,2,1,8 16:1,3,1,6 16:1,6,1,3 16:2,3,1,6 4:4,4,1,4 4:2,4,2,4 4:4,2,1,8 4:2,2,1,8 4:3,4,1,4 4:4,4,2,4 4:6,2,1,8 4:2,4,1,4 > $O/mcnull_grids.log 2>&1); grep "^mcnull" $O/mcnull_grids.log | cut -c1-260
