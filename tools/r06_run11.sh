mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python tools/k2c_ab.py cfg2:10000 -- default compress_theta=0.4 compress_theta=0.6 compress_theta=0.7 compress_theta=0.85 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/theta_sweep_gemm.txt
timeout 900 python tools/k2c_ab.py cfg3:100000 cfg4:62464 -- default compress_theta=0.6 compress_theta=0.8 compress_theta=0.9 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06/theta_sweep_gemm.txt
