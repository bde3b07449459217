mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python tools/k2c_ab.py cfg2:10000 test1 cfg3:100000 -- k1_balance=0 k1_balance=1 k1_balance=1,k2_skip_epilogue=1 2>&1 | tee gpurun_out/r06/k1_balance_epilogue_ab.txt | tail -20
for t in cfg2:10000 test1; do
  timeout 300 python tools/cold_evals.py $t 2>&1 | tee -a gpurun_out/r06/cold_evals.txt
  timeout 300 python tools/cold_evals.py $t k2tune=0 2>&1 | tee -a gpurun_out/r06/cold_evals.txt
done
