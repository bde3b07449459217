mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1200 python tools/k2c_ab.py cfg3:100000 cfg4:62464 cfg5:100000 cfg2:40000 cfg2:160000 -- default compress_theta=0.9 compress_theta=0.95 compress_theta=1.0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/theta_sweep_gemm2.txt
