mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06/tests_full.txt
cat gpurun_out/r06/tests_full.txt
timeout 600 python tools/k2c_ab.py cfg2:10000 test1 cfg3:100000 cfg4:62464 cfg5:100000 -- k2c_gemm=0 default 2>&1 | tee gpurun_out/r06/k2c_ab2.txt | tail -30
