mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python tools/k2c_ab.py cfg2:10000 test1 cfg3:100000 cfg4:62464 -- k2_epilogue=wave k2_epilogue=lane 2>&1 | tee gpurun_out/r06/epilogue_ab.txt | tail -20
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06/tests_full2.txt
cat gpurun_out/r06/tests_full2.txt
