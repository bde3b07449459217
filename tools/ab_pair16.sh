# (needs tools/_variants/pair16 = the source with profiles/r05/walk16_paired_tiles.patch applied, built with -DCAFE_K2_PAIR16=1)
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
CAFEHIP_LIB=tools/_variants/pair16/libcafehip.so timeout 900 python -m pytest tests/test_gpu_compression.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
bash tools/ab_bench.sh cafe_amd/lib/libcafehip.so tools/_variants/pair16/libcafehip.so cfg3 cfg4 cfg5
