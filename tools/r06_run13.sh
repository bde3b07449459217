mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1200 python tools/k2c_ab.py cfg3:100000 cfg3:40000 cfg3:20000 cfg5:100000 -- compress_theta=0.7 default 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/theta_sweep_gemm3.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -4
