#!/bin/bash
# Round 6, GPU session 5: batch JPEG decode with one page-locked staging buffer and chunked copies (virtex_amd/jpeg.py) -- the GPU
# tests of the decoder and its throughput per host thread count.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_jpeg.py tests/test_data.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_s5_tests.txt
timeout 400 python tools/bench_jpeg.py > gpurun_out/r06_s5_bench_jpeg.txt 2>&1
timeout 400 python tools/bench_jpeg.py --images 1024 > gpurun_out/r06_s5_bench_jpeg_1024.txt 2>&1
cat gpurun_out/r06_s5_tests.txt gpurun_out/r06_s5_bench_jpeg.txt gpurun_out/r06_s5_bench_jpeg_1024.txt
