#!/bin/bash
# round 3, call al: PMC traffic of every workload and the cfg3 / cfg5 bench lines re-taken after the trimmed park fetch
mkdir -p gpurun_out/r03al; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03al
(timeout 2400 python tools/collect_pmc.py $O/pmc > $O/pmc.log 2>&1; echo "rc=$?" >> $O/pmc.log); tail -7 $O/pmc.log | cut -c1-250
cp $O/pmc/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
for c in cfg3 cfg5; do (timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err); head -c 250 $O/bench_$c.json; echo; done
rm -rf $O/pmc/*_SIZE
