mkdir -p gpurun_out/final2; export TMPDIR=/tmp; O=gpurun_out/final2
(timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_driver_sized.json 2> $O/bench_cfg2_driver.err; echo "rc=$?" >> $O/bench_cfg2_driver.err); head -c 200 $O/bench_cfg2_driver_sized.json; echo
(timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?" >> $O/bench_cfg2.err); head -c 200 $O/bench_cfg2.json; echo
for c in cfg3 cfg4 cfg5; do
  (timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err); head -c 200 $O/bench_$c.json; echo
done
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -3
