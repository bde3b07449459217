mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python tools/k2c_ab.py cfg2:10000 test1 cfg3:100000 cfg4:62464 -- k2_epilogue=wave k2_epilogue=lane 2>&1 | tee gpurun_out/r06/epilogue_ab2.txt | tail -20
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_compression.py tests/test_gpu_multi_eval.py -x -q 2>&1 | tail -4
rm -f gpurun_out/r06/cold_evals2.txt
for t in cfg2:10000 test1; do
  timeout 300 python tools/cold_evals.py $t 2>&1 | tee -a gpurun_out/r06/cold_evals2.txt | head -12
  CAFEHIP_PRELOAD=0 timeout 300 python tools/cold_evals.py $t 2>&1 | tee -a gpurun_out/r06/cold_evals2.txt | head -4
done
