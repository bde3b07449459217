export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_compression.py tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_matrix_cache.py -x -q 2>&1 | tail -8
