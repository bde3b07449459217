mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python tools/k2c_ab.py cfg2:10000 test1 cfg4:62464 -- k1_balance=0 k1_balance=11 k1_balance=2 k1_balance=3 k1_balance=0,k1kpb=2 2>&1 | tee gpurun_out/r06/k1_order_ab.txt | tail -24
