mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_compression.py tests/test_gpu_fuzz.py tests/test_gpu_matrix_cache.py -x -q 2>&1 | tail -5 > gpurun_out/r06/tests1.txt
cat gpurun_out/r06/tests1.txt
timeout 1200 python tools/k2c_ab.py cfg2:10000 test1 cfg3:100000 cfg4:62464 -- k2c_gemm=0 k2c_gemm=1 k2c_gemm=1,k2c_nst=1 k2c_gemm=1,k2c_nst=2 k2c_gemm=1,k2c_nst=4 k2c_gemm=1,k2c_nst=4,k2c_pair=1 k2c_gemm=1,k2c_nst=2,k2c_pair=1 k2c_gemm=1,k2c_nst=4,k2c_pair=0 k2c_gemm=1,k2c_xcd=0 2>&1 | tee gpurun_out/r06/k2c_ab1.txt | tail -60
