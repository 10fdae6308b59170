#!/bin/bash
# Round 4, GPU session 3: lean interior-tile statistics epilogues -- stamps of the generation-3 kernels, the step itself
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
VIRTEX_AMD_LIB=$R/virtex_amd/lib/libvirtex_amd_ablate.so timeout 600 python tools/ablate_gen3.py --stats > gpurun_out/r04_s3_stats_epilogues.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 30 --warmup 10 > gpurun_out/r04_s3_bench.json 2> gpurun_out/r04_s3_bench.err
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r04_s3_tests.txt
grep -E "cand|epilogue" gpurun_out/r04_s3_stats_epilogues.txt | head -80
cat gpurun_out/r04_s3_bench.json | head -c 600; echo; tail -3 gpurun_out/r04_s3_bench.err; cat gpurun_out/r04_s3_tests.txt
