mkdir -p gpurun_out/r06
export TMPDIR=/tmp
for pass in 1 2; do
for v in main g2 g8; do
  if [ $v = main ]; then unset CAFEHIP_LIB; else export CAFEHIP_LIB=tools/_variants/$v/libcafehip.so; fi
  echo "### ring depth variant $v pass $pass"
  timeout 600 python tools/k2c_ab.py cfg2:10000 test1 cfg3:100000 cfg4:62464 -- default 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r06/k2c_gemm_depth_ab.txt
